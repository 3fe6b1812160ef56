#!/bin/bash
set +e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05_c5_gputest.txt
python tools/resnet_profile.py 32 20 train 2>&1 | grep -v amdgpu.ids | tail -1 > gpurun_out/r05_c5_resnet.txt
for w in 4 8; do for b in 1 4; do
  TELL_SK_WAVES=$w python bench.py --generate --beam $b 2>/dev/null | tail -1 > gpurun_out/r05_c5_gen_w${w}_beam$b.json
done; done
python bench.py --no-cpu-baseline 2> gpurun_out/r05_c5_bench.err | tail -1 > gpurun_out/r05_c5_bench.json
