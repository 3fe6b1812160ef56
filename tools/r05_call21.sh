#!/bin/bash
# gemm_s64.hip (TELL_GEMM_S64=1/2/3): correctness + per-shape timing + trunk + decoder GEMMs
set +e
mkdir -p gpurun_out
: > gpurun_out/r05_c21_tests.txt
for m in 1 2; do
TELL_GEMM_S64=$m timeout 900 python -m pytest tests/test_gpu_encoders.py -x -q -k "resnet" 2>&1 | tail -3 >> gpurun_out/r05_c21_tests.txt
TELL_GEMM_S64=$m timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm or linear" 2>&1 | tail -3 >> gpurun_out/r05_c21_tests.txt
done
timeout 600 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_c21_conv.txt
for ks in 0 1 2 3 0 1 2 3; do
  echo "TELL_GEMM_S64=$ks" >> gpurun_out/r05_c21_conv.txt
  TELL_GEMM_S64=$ks timeout 300 python tools/resnet_profile.py 32 20 train 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r05_c21_conv.txt
done
for ks in 0 1 2; do
  echo "TELL_GEMM_S64=$ks" >> gpurun_out/r05_c21_conv.txt
  TELL_GEMM_S64=$ks timeout 300 python tools/bench_decoder_gemms.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_c21_conv.txt
done
