#!/bin/bash
# round 4, call 1: q4 kernel first light + baseline numbers of this box
set +e
mkdir -p gpurun_out
echo "== q4 check (default dispatch)"; TELL_GEMM_Q4=1 timeout 300 python tools/probes/q4_check.py 2>&1 | grep -v amdgpu.ids | tail -16
echo "== q4 check (forced, partial rounds)"; TELL_GEMM_Q4=1 TELL_GEMM_TILE=8 timeout 300 python tools/probes/q4_check.py 2>&1 | grep -v amdgpu.ids | tail -16
echo "== roberta gemms pp2"; timeout 300 python tools/bench_roberta_gemms.py 2>&1 | grep -v amdgpu.ids | tail -6
echo "== roberta gemms q4"; TELL_GEMM_Q4=1 timeout 300 python tools/bench_roberta_gemms.py 2>&1 | grep -v amdgpu.ids | tail -6
echo "== roberta gemms pp2 (again)"; timeout 300 python tools/bench_roberta_gemms.py 2>&1 | grep -v amdgpu.ids | tail -6
echo "== roberta gemms q4 (again)"; TELL_GEMM_Q4=1 timeout 300 python tools/bench_roberta_gemms.py 2>&1 | grep -v amdgpu.ids | tail -6
echo "== pmc LDS counters pp2"
bash tools/pmc_kernel.sh gemm_nt_pp2 gpurun_out/r04_pmc_gemm_lds_pp2.txt SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_INSTS_LDS,SQ_INST_CYCLES_VMEM,SQ_WAVE_CYCLES,SQ_WAIT_INST_LDS,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES -- python tools/bench_roberta_gemms.py > /dev/null
cat gpurun_out/r04_pmc_gemm_lds_pp2.txt
echo "== pmc LDS counters q4"
TELL_GEMM_Q4=1 bash tools/pmc_kernel.sh gemm_nt_q4 gpurun_out/r04_pmc_gemm_lds_q4.txt SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_INSTS_LDS,SQ_INST_CYCLES_VMEM,SQ_WAVE_CYCLES,SQ_WAIT_INST_LDS,SQ_VALU_MFMA_BUSY_CYCLES,SQ_BUSY_CYCLES -- python tools/bench_roberta_gemms.py > /dev/null
cat gpurun_out/r04_pmc_gemm_lds_q4.txt
echo "== bench (baseline, default kernels)"; timeout 600 python bench.py 2> gpurun_out/r04_bench0.err | tail -1 > gpurun_out/r04_bench0.json; cut -c1-600 gpurun_out/r04_bench0.json
