#!/bin/bash
set +e
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "q4_kernel" 2>&1 | tail -3
run() { echo "== bench $*"; env "$@" timeout 900 python bench.py --no-cpu-baseline --no-secondary --no-generation --no-dp-selftest 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); v=d['loader_variable_lengths']; print(d['value'], d['loader']['value'], v['value'], v['first_epochs_value'], v['step_signatures_seen'])"; }
for rep in 1 2; do
run TELL_Q4_PARTIAL=0
run TELL_Q4_PARTIAL=1
done
