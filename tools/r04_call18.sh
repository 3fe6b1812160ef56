#!/bin/bash
set +e
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "dynconv or dynamic_conv or glu or generation_step" 2>&1 | tail -3
timeout 300 python tools/bench_dynconv.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_dynconv.txt | tail -5
for v in 0 1 0 1; do
  echo "== decoder alone TELL_DYNCONV_BLOCK=$v"
  TELL_DYNCONV_BLOCK=$v timeout 300 python tools/decoder_profile.py faces_objects 32 20 2>&1 | tail -2 | head -1
done
timeout 1500 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_train.py tests/test_gpu_fullsize.py -q -x 2>&1 | tail -4
