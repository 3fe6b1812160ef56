#!/bin/bash
# training step with the small-kernel streams confined to a CU range (q4 on per-XCD tile counters)
mkdir -p gpurun_out
: > gpurun_out/train_part.txt
run() {
  echo "## $1" >> gpurun_out/train_part.txt
  env $1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-generation --no-loader --no-dp-selftest --no-roofline 2>/dev/null | tail -1 | python -c "
import sys, json
j = json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'])" >> gpurun_out/train_part.txt
}
run "X=0"
run "TELL_Q4_DYNAMIC=1"
run "TELL_Q4_DYNAMIC=1 TELL_SIDE_CUS=resnet=0-64"
run "TELL_Q4_DYNAMIC=1 TELL_SIDE_CUS=resnet=0-128"
run "TELL_Q4_DYNAMIC=1 TELL_SIDE_CUS=resnet=0-128 TELL_MAIN_CUS=0-128"
run "TELL_Q4_DYNAMIC=1 TELL_SIDE_CUS=resnet=0-96 TELL_MAIN_CUS=96-192"
run "TELL_Q4_DYNAMIC=1 TELL_MAIN_CUS=0-128"
run "TELL_SIDE_CUS=resnet=0-128"
run "X=0"
