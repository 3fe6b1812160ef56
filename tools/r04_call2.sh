#!/bin/bash
set +e
for v in 0 1 2; do echo "== q4 check VAR=$v"; TELL_GEMM_Q4=1 TELL_Q4_VAR=$v timeout 300 python tools/probes/q4_check.py 2>&1 | grep -v amdgpu.ids | tail -3; done
for rep in 1 2; do
for v in 0 1 2; do echo "== roberta gemms q4 VAR=$v (rep $rep)"; TELL_GEMM_Q4=1 TELL_Q4_VAR=$v timeout 300 python tools/bench_roberta_gemms.py 2>&1 | grep -v amdgpu.ids | tail -5; done
done
for a in 1 2; do echo "== roberta gemms q4 ABL=$a"; TELL_GEMM_Q4=1 TELL_Q4_ABL=$a timeout 300 python tools/bench_roberta_gemms.py 2>&1 | grep -v amdgpu.ids | tail -5; done
echo "== pp2"; timeout 300 python tools/bench_roberta_gemms.py 2>&1 | grep -v amdgpu.ids | tail -5
