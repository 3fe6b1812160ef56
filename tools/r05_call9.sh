#!/bin/bash
set +e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_decoder.py -x -q 2>&1 | tail -5 > gpurun_out/r05_c9_tests.txt
for g in 0 1 0 1; do for b in 1 4; do
  TELL_HEAD_GROUPED=$g python bench.py --generate --beam $b 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('HEAD_GROUPED=$g beam $b', d['value'], d['roofline']['avg_step_us'], d['roofline']['frac'])" >> gpurun_out/r05_c9_head.txt
done; done
