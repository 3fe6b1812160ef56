#!/bin/bash
# K-split form of the 64x64 direct-to-LDS kernel (TELL_GEMM_KS=1): correctness + per-shape timing + trunk + decoder GEMMs
set +e
mkdir -p gpurun_out
TELL_GEMM_KS=1 timeout 900 python -m pytest tests/test_gpu_encoders.py -x -q -k "resnet" 2>&1 | tail -3 > gpurun_out/r05_c19_tests.txt
TELL_GEMM_KS=1 timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm" 2>&1 | tail -3 >> gpurun_out/r05_c19_tests.txt
timeout 600 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_c19_conv.txt
for ks in 0 1 0 1; do
  echo "TELL_GEMM_KS=$ks" >> gpurun_out/r05_c19_conv.txt
  TELL_GEMM_KS=$ks timeout 300 python tools/resnet_profile.py 32 20 train 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r05_c19_conv.txt
done
for ks in 0 1; do
  echo "TELL_GEMM_KS=$ks" >> gpurun_out/r05_c19_conv.txt
  TELL_GEMM_KS=$ks timeout 300 python tools/bench_decoder_gemms.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_c19_conv.txt
done
