#!/bin/bash
set +e
mkdir -p gpurun_out
for r in 0 128 32 0; do
  TELL_SK_ROWS=$r python bench.py --generate --beam 4 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('SK_ROWS=$r beam 4', d['value'], d['roofline']['avg_step_us'], d['roofline']['frac'])" >> gpurun_out/r05_c8_skrows.txt
done
