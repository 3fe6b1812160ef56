#!/bin/bash
set +e
for v in 0 256 0 256; do
  echo "== decoder alone TELL_KV_PITCH_PAD=$v"
  TELL_KV_PITCH_PAD=$v timeout 300 python tools/decoder_profile.py faces_objects 32 20 2>&1 | tail -2 | head -1
done
timeout 1500 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_train.py tests/test_gpu_fullsize.py tests/test_gpu_step_graph.py -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
run() { echo "== bench $*"; env "$@" timeout 900 python bench.py --no-cpu-baseline --no-secondary --no-generation --no-dp-selftest --no-loader 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['decoder_step']['alone_ms'])"; }
for rep in 1 2; do
run TELL_KV_PITCH_PAD=0
run TELL_KV_PITCH_PAD=256
done
