#!/bin/bash
set +e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_ops.py tests/test_gpu_fullsize.py tests/test_gpu_train.py -x -q 2>&1 | tail -5 > gpurun_out/r05_c11_tests.txt
for i in 1 2; do for b in 1 4; do
  python bench.py --generate --beam $b 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('beam $b', d['value'], d['roofline']['avg_step_us'], d['roofline']['frac'])" >> gpurun_out/r05_c11_gen.txt
done; done
python tools/decoder_profile.py faces_objects 32 30 2>&1 | grep -v amdgpu.ids | head -1 > gpurun_out/r05_c11_decoder.txt
SEQ_ANCHOR=greedy_update bash tools/profile_cmd.sh r05_c11_generate "greedy generation" python bench.py --generate --beam 1 --steps 1 --warmup 1
bash tools/profile_cmd.sh r05_c11_decoder "decoder half alone" python tools/decoder_profile.py faces_objects 32 20
