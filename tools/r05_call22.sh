#!/bin/bash
set +e
mkdir -p gpurun_out
TELL_GEMM_S64=1 timeout 900 python -m pytest tests/test_gpu_encoders.py -x -q -k "first_blocks" 2>&1 | tail -40 > gpurun_out/r05_c22_tests.txt
TELL_GEMM_S64=2 timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "grouped_linear" 2>&1 | tail -40 >> gpurun_out/r05_c22_tests.txt
