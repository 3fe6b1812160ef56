#!/bin/bash
set +e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py -x -q -k "gemm" 2>&1 | grep -E "passed|failed" > gpurun_out/r05_c15_tests.txt
python -m pytest tests/test_gpu_encoders.py -x -q -k "roberta" 2>&1 | grep -E "passed|failed" >> gpurun_out/r05_c15_tests.txt
python tools/bench_roberta_gemms.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_c15_gemms.txt
python tools/bench_roberta_gemms.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_c15_gemms.txt
