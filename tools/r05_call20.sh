#!/bin/bash
set +e
mkdir -p gpurun_out
: > gpurun_out/r05_c20_pmc.txt
for shape in "14 256 3 1 256" "14 1024 1 1 256" "7 512 3 1 512"; do
  for ctr in FETCH_SIZE TCC_HIT_sum,TCC_MISS_sum SQ_BUSY_CU_CYCLES,SQ_WAIT_INST_LDS,SQ_INSTS_LDS,SQ_WAVE_CYCLES,SQ_WAIT_ANY,SQ_ACTIVE_INST_LDS,SQ_LDS_BANK_CONFLICT; do
    echo "## conv_one $shape  [$ctr]" >> gpurun_out/r05_c20_pmc.txt
    bash tools/pmc_kernel.sh gemm_nt_glds /tmp/o.txt $ctr -- python tools/probes/conv_one.py $shape >> gpurun_out/r05_c20_pmc.txt 2>&1
    grep "^M " /tmp/pmck.log >> gpurun_out/r05_c20_pmc.txt
  done
done
