#!/bin/bash
# q4 as the default: GPU test suite + same-box bench A/B
set +e
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r04_gputest_a.txt; tail -6 gpurun_out/r04_gputest_a.txt
for rep in 1 2; do
  echo "== bench q4 default (rep $rep)"; timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-generation --no-loader 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('achieved'), d.get('roofline',{}).get('isolated'))"
  echo "== bench pp2 (rep $rep)"; TELL_GEMM_Q4=0 timeout 600 python bench.py --no-cpu-baseline --no-secondary --no-generation --no-loader 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('roofline',{}).get('achieved'), d.get('roofline',{}).get('isolated'))"
done
