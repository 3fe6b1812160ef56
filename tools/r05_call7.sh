#!/bin/bash
set +e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_step_graph.py tests/test_gpu_ops.py -x -q 2>&1 | tail -6 > gpurun_out/r05_c7_tests.txt
for k in 4096 1024 2048 4096 1024; do
  echo "TELL_GROUP_WIDE_K=$k" >> gpurun_out/r05_c7_wide.txt
  TELL_GROUP_WIDE_K=$k python tools/decoder_profile.py faces_objects 32 30 2>&1 | grep -v amdgpu.ids | head -1 >> gpurun_out/r05_c7_wide.txt
done
SEQ_ANCHOR=bertadam bash tools/profile_cmd.sh r05_c7_decoder "decoder half alone" python tools/decoder_profile.py faces_objects 32 20
TELL_GROUP_WIDE_K=1024 bash tools/profile_cmd.sh r05_c7_decoder_wide "decoder half alone, TELL_GROUP_WIDE_K=1024" python tools/decoder_profile.py faces_objects 32 20
