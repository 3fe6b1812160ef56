#!/bin/bash
set +e
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_encoders.py -x -q -k "roberta_large_all_24 or resnet" -s 2>&1 | grep -E "passed|failed|RoBERTa-large, 24" > gpurun_out/r05_c25_tests.txt
timeout 900 python -m pytest tests/test_gpu_ops.py -x -q -k "gemm or linear" 2>&1 | tail -2 >> gpurun_out/r05_c25_tests.txt
: > gpurun_out/r05_c25.txt
for d in 0 1 0 1; do
  echo "TELL_S64_DEEP=$d" >> gpurun_out/r05_c25.txt
  TELL_S64_DEEP=$d timeout 300 python tools/decoder_profile.py faces_objects 32 30 2>&1 | grep -v amdgpu.ids | tail -2 | head -1 >> gpurun_out/r05_c25.txt
  TELL_S64_DEEP=$d timeout 300 python tools/resnet_profile.py 32 20 train 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r05_c25.txt
done
for d in 0 1; do
  echo "TELL_S64_DEEP=$d" >> gpurun_out/r05_c25.txt
  TELL_S64_DEEP=$d timeout 300 python tools/bench_decoder_gemms.py 2>&1 | grep -v amdgpu.ids | head -6 >> gpurun_out/r05_c25.txt
done
TELL_S64_DEEP=1 SEQ_ANCHOR=bertadam_update bash tools/profile_cmd.sh r05_c25_dec_deep "decoder alone, s64 deep" python tools/decoder_profile.py faces_objects 32 20
