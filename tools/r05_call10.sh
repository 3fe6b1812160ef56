#!/bin/bash
set +e
mkdir -p gpurun_out
for v in "0 0" "1 0" "0 1" "1 1" "0 0" "1 1"; do set -- $v; for b in 1 4; do
  TELL_SK_NT=$1 TELL_AD_NT=$2 python bench.py --generate --beam $b 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('SK_NT=$1 AD_NT=$2 beam $b', d['value'], d['roofline']['avg_step_us'], d['roofline']['frac'])" >> gpurun_out/r05_c10_nt.txt
done; done
