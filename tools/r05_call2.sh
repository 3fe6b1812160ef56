#!/bin/bash
# round 5, GPU call 2: suite + deep-ring sweeps (conv shapes, decoder GEMM shapes) + A/B of the deep ring in ResNet / decoder half
set +e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r05_c2_gputest.txt
python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_c2_conv.txt
for d in 0 1; do
  echo "TELL_GEMM_DEEP=$d" >> gpurun_out/r05_c2_decgemms.txt
  TELL_GEMM_DEEP=$d python tools/bench_decoder_gemms.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_c2_decgemms.txt
  echo "TELL_CONV_DEEP=$d" >> gpurun_out/r05_c2_resnet.txt
  TELL_CONV_DEEP=$d python tools/resnet_profile.py 32 20 train 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r05_c2_resnet.txt
  TELL_CONV_DEEP=$d python tools/resnet_profile.py 32 20 eval 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r05_c2_resnet.txt
  echo "TELL_GEMM_DEEP=$d" >> gpurun_out/r05_c2_decoder.txt
  TELL_GEMM_DEEP=$d python tools/decoder_profile.py faces_objects 32 30 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_c2_decoder.txt
done
for d in "0 0" "1 1"; do
  set -- $d
  TELL_GEMM_DEEP=$1 TELL_CONV_DEEP=$2 python bench.py --no-cpu-baseline --no-secondary --no-generation --no-loader --no-dp-selftest --no-many-signatures 2>/dev/null | tail -1 > gpurun_out/r05_c2_bench_deep$1.json
done
