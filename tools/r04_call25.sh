#!/bin/bash
set +e
timeout 3000 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
NEW=transform-and-tell_amd/csrc/libtell_hip.so
cp $NEW _ab/libtell_new.so
for rep in 1 2; do
  for which in old new; do
    cp _ab/libtell_$which.so $NEW
    echo "== $which"; timeout 300 python tools/decoder_profile.py faces_objects 32 20 2>&1 | tail -2 | head -1
    timeout 300 python tools/resnet_profile.py 32 20 train 2>&1 | tail -1
  done
done
for which in old new; do
  cp _ab/libtell_$which.so $NEW
  echo "== bench $which"; timeout 900 python bench.py --no-cpu-baseline --no-secondary --no-generation --no-dp-selftest --no-loader 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['decoder_step']['alone_ms'])"
  timeout 300 python bench.py --generate --beam 1 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('greedy', d['value'], d['roofline']['avg_step_us'])"
done
cp _ab/libtell_new.so $NEW
