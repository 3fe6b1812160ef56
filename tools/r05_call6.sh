#!/bin/bash
set +e
mkdir -p gpurun_out
python -m pytest tests/test_gpu_step_graph.py -x -q -k "survive" 2>&1 | tail -45 > gpurun_out/r05_c6_failtest.txt
python -m pytest tests -m gpu -q 2>&1 | tail -12 > gpurun_out/r05_c6_gputest.txt
for v in "0 0" "4 1" "0 0" "4 1"; do
  set -- $v
  TELL_ADAM_VAR=$1 TELL_LN_VAR=$2 python bench.py --no-cpu-baseline --no-secondary --no-generation --no-loader --no-dp-selftest --no-many-signatures --no-pmc 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('ADAM_VAR=$1 LN_VAR=$2', d['value'], d['ms_per_step'], d['decoder_step']['alone_ms'])" >> gpurun_out/r05_c6_ab.txt
done
for f in 0 1; do for b in 1 4; do
  TELL_DECODE_FOLD=$f python bench.py --generate --beam $b 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('FOLD=$f beam $b', d['value'], d['roofline']['avg_step_us'], d['roofline']['frac'])" >> gpurun_out/r05_c6_gen.txt
done; done
