#!/bin/bash
set +e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r05_c4_gputest.txt
python tools/bench_layernorm.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_c4_ln.txt
for v in "0" "1"; do
  echo "TELL_BN_FA=$v" >> gpurun_out/r05_c4_resnet.txt
  TELL_BN_FA=$v python tools/resnet_profile.py 32 20 train 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r05_c4_resnet.txt
done
for f in 0 1; do for b in 1 4; do
  TELL_DECODE_FOLD=$f python bench.py --generate --beam $b 2>/dev/null | tail -1 > gpurun_out/r05_c4_gen_fold${f}_beam$b.json
done; done
