#!/bin/bash
set +e
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_decoder.py tests/test_gpu_data_eval.py -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5 > gpurun_out/gen_large.txt
for cfg in "128 4" "64 4" "256 1" "32 4"; do
  set -- $cfg
  echo "# B=$1 beam=$2" >> gpurun_out/gen_large.txt
  timeout 600 python bench.py --generate --batch $1 --beam $2 2>&1 | grep -v amdgpu.ids | tail -1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); print(json.dumps({k: j[k] for k in ('value', 'serial_value', 'ms_per_step')}), j['roofline']['avg_step_us'], j['roofline']['frac'])
    else:
        print(ln.rstrip()[-300:])
" >> gpurun_out/gen_large.txt
done
