#!/bin/bash
set +e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_step_graph.py -x -q -k "survive" 2>&1 | tail -40 > gpurun_out/r05_c3_failtest.txt
python -m pytest tests/test_gpu_encoders.py -x -q 2>&1 | tail -5 > gpurun_out/r05_c3_enctest.txt
for v in "0 0" "1 0" "1 1" "0 1"; do
  set -- $v
  echo "TELL_BN_FA=$1 TELL_BN_COMBINE=$2" >> gpurun_out/r05_c3_resnet.txt
  TELL_BN_FA=$1 TELL_BN_COMBINE=$2 python tools/resnet_profile.py 32 20 train 2>&1 | grep -v amdgpu.ids | tail -1 >> gpurun_out/r05_c3_resnet.txt
done
TELL_BN_FA=0 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_c3_conv_old.txt
TELL_BN_FA=1 python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_c3_conv_new.txt
