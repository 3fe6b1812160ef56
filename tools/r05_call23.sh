#!/bin/bash
# gemm_s64.hip as the default from K = 1024: tests, conv shapes, trunk, decoder GEMMs, decoder step, bench line
set +e
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_encoders.py -x -q 2>&1 | tail -3 > gpurun_out/r05_c23_tests.txt
timeout 1200 python -m pytest tests/test_gpu_ops.py -x -q 2>&1 | tail -3 >> gpurun_out/r05_c23_tests.txt
timeout 1200 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3 >> gpurun_out/r05_c23_tests.txt
timeout 600 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_c23_conv.txt
for ks in 0 1 0 1; do
  echo "TELL_GEMM_S64=$ks" >> gpurun_out/r05_c23_conv.txt
  TELL_GEMM_S64=$ks timeout 300 python tools/resnet_profile.py 32 20 train 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r05_c23_conv.txt
  TELL_GEMM_S64=$ks timeout 300 python tools/resnet_profile.py 32 20 eval 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r05_c23_conv.txt
done
for ks in 0 1; do
  echo "TELL_GEMM_S64=$ks" >> gpurun_out/r05_c23_conv.txt
  TELL_GEMM_S64=$ks timeout 300 python tools/bench_decoder_gemms.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_c23_conv.txt
  TELL_GEMM_S64=$ks timeout 300 python tools/decoder_profile.py faces_objects 32 30 2>&1 | grep -v amdgpu.ids | tail -2 >> gpurun_out/r05_c23_conv.txt
done
for ks in 0 1 0 1; do
  TELL_GEMM_S64=$ks timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-generation --no-loader --no-dp-selftest 2>&1 | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print('S64=$ks', j['value'], 'samples/s', j['ms_per_step'], 'ms; decoder alone', j['decoder_step']['alone_ms'])" >> gpurun_out/r05_c23_conv.txt
done
