#!/bin/bash
# PMC of layer3's 3x3 convolution (B = 32): the general direct-to-LDS body against gemm_s64.hip
set +e
mkdir -p gpurun_out
: > gpurun_out/r05_pmc_conv_s64.txt
for m in 0 1; do
  for ctr in SQ_INSTS_VALU,SQ_INSTS_SALU,SQ_INSTS_LDS,SQ_INSTS_VMEM_RD,SQ_INSTS_MFMA,SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES,SQ_ACTIVE_INST_ANY,SQ_INST_CYCLES_SALU,SQ_WAIT_INST_LDS,SQ_LDS_BANK_CONFLICT TCC_HIT_sum,TCC_MISS_sum; do
    echo "## TELL_GEMM_S64=$m conv_one 14 256 3 1 256  [$ctr]" >> gpurun_out/r05_pmc_conv_s64.txt
    TELL_GEMM_S64=$m bash tools/pmc_kernel.sh gemm_nt_ /tmp/o.txt $ctr -- python tools/probes/conv_one.py 14 256 3 1 256 >> gpurun_out/r05_pmc_conv_s64.txt 2>&1
  done
done
