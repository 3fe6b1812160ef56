#!/bin/bash
# PMC passes over the generation step's one-query attention (attn_decode_kernel), beam $1 -> gpurun_out/$2
beam=${1:-4}; out=gpurun_out/${2:-r06_pmc_attn_decode.txt}
: > $out
for ctr in SQ_WAVES,SQ_BUSY_CYCLES,SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_VALU,SQ_INST_CYCLES_VMEM,SQ_ACTIVE_INST_ANY TCC_HIT_sum,TCC_MISS_sum TCC_EA_RDREQ_sum,TCC_EA_RDREQ_32B_sum TCP_TCC_READ_REQ_sum,TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum,TCP_TCP_LATENCY_sum TA_BUSY_avr,TA_TA_BUSY_sum GRBM_GUI_ACTIVE,TCC_BUSY_avr; do
  echo "## beam $beam [$ctr]" >> $out
  bash tools/pmc_kernel.sh attn_decode /tmp/o.txt $ctr -- python bench.py --generate --beam $beam --steps 1 --warmup 1 >> $out 2>&1
done
