#!/bin/bash
set +e
timeout 600 python -m pytest tests/test_gpu_ops.py -q -k "q4 or pingpong or variants" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_fullsize.py -q -k "early_eos or beam4_matches" 2>&1 | tail -12
run() { echo "== bench $*"; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-generation --no-loader 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('achieved'), d.get('roofline',{}).get('avg_launch_us'), d.get('roofline',{}).get('isolated'))"; }
for rep in 1 2; do
run TELL_Q4_DYNAMIC=0 TELL_GEMM_Q4E=1
run TELL_Q4_DYNAMIC=1 TELL_GEMM_Q4E=1
run TELL_Q4_DYNAMIC=0 TELL_GEMM_Q4E=0
run TELL_Q4_DYNAMIC=0 TELL_GEMM_Q4E=2
run TELL_GEMM_Q4=0
done
