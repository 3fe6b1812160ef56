#!/bin/bash
set +e
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "residual or q4" 2>&1 | tail -4
timeout 300 python tools/probes/q4_residual.py 2>&1 | tail -8
timeout 600 python -m pytest tests/test_gpu_encoders.py -q -k "roberta" 2>&1 | tail -4
run() { echo "== bench $*"; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-generation --no-loader --no-dp-selftest 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('achieved'), d.get('roofline',{}).get('avg_launch_us'), d.get('roofline',{}).get('isolated'))"; }
for rep in 1 2; do
run TELL_GEMM_RESIDUAL=0
run TELL_GEMM_RESIDUAL=1
done
