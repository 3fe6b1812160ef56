#!/bin/bash
set +e
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "dynconv_block or dynamic_conv" 2>&1 | tail -8
timeout 300 python tools/bench_dynconv.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_dynconv.txt
for v in 0 1; do
  echo "== decoder alone TELL_DYNCONV_BLOCK=$v"
  TELL_DYNCONV_BLOCK=$v timeout 300 python tools/decoder_profile.py faces_objects 32 20 2>&1 | tail -2
done
timeout 900 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_train.py -q -x 2>&1 | tail -4
