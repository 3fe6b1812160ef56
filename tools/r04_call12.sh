#!/bin/bash
set +e
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "split_k or wn or adam or optim" 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_gpu_step_graph.py tests/test_gpu_train.py -q -x 2>&1 | tail -6
for gs in 0 1; do
  for sk in 0 1; do
    echo "== decoder alone TELL_GRAD_STORE=$gs SPLITK=$sk"
    TELL_GRAD_STORE=$gs TELL_GEMM_SKINNY_SPLITK=$sk timeout 300 python tools/decoder_profile.py faces_objects 32 20 2>&1 | tail -2
  done
done
run() { echo "== bench $*"; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-generation --no-loader --no-dp-selftest 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('achieved'), d.get('roofline',{}).get('avg_launch_us'), d.get('decoder_step'))"; }
for rep in 1 2; do
run TELL_GRAD_STORE=0 TELL_GEMM_SKINNY_SPLITK=0
run TELL_GRAD_STORE=1 TELL_GEMM_SKINNY_SPLITK=1
done
timeout 200 python tools/probes/q4_residual.py "res q4" 2>&1 | tail -3
timeout 200 python tools/probes/q4_residual.py "res pp2" 2>&1 | tail -3
timeout 1800 python -m pytest tests/test_gpu_fullsize.py -q -x 2>&1 | tail -6
