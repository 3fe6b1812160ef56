#!/bin/bash
# PMC passes over the generation step's skinny linears, beam $1 -> gpurun_out/$2
beam=${1:-4}; out=gpurun_out/${2:-r06_pmc_skinny.txt}
: > $out
for ctr in SQ_WAVES,SQ_BUSY_CYCLES,SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_INST_CYCLES_VMEM,SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE,TCC_BUSY_avr TCP_TCC_READ_REQ_sum,TCP_TCC_READ_REQ_LATENCY_sum TCC_HIT_sum,TCC_MISS_sum SQ_VALU_MFMA_BUSY_CYCLES,SQ_ACTIVE_INST_LDS,SQ_INSTS_VALU,SQ_INSTS_VMEM_RD,SQ_INSTS_SALU,SQ_INSTS_SMEM,SQ_INSTS_LDS; do
  echo "## beam $beam [$ctr]" >> $out
  bash tools/pmc_kernel.sh skinny_mfma /tmp/o.txt $ctr -- python bench.py --generate --beam $beam --steps 1 --warmup 1 >> $out 2>&1
done
