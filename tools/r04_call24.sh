#!/bin/bash
set +e
for v in 0 128 0 128; do
  echo "== decoder alone TELL_KV_PITCH_PAD1=$v"
  TELL_KV_PITCH_PAD1=$v timeout 300 python tools/decoder_profile.py faces_objects 32 20 2>&1 | tail -2 | head -1
done
run() { echo "== bench $*"; env "$@" timeout 900 python bench.py --no-cpu-baseline --no-secondary --no-generation --no-dp-selftest --no-loader 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['decoder_step']['alone_ms'])"; }
for rep in 1 2; do
run TELL_QKV_PAD=0
run TELL_QKV_PAD=256
done
