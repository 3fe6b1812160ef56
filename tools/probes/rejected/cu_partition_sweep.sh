#!/bin/bash
# generation with the decode chain on its own CUs (streams.CUPartition): decode_cus sweep, batch 32 / 128, greedy / beam 4
set +e
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_data_eval.py -x -q 2>&1 | tail -5 > gpurun_out/r05_c17_tests.txt
: > gpurun_out/r05_c17_gen.txt
for cfg in "32 1 0" "32 1 32" "32 1 64" "32 1 128" "32 4 64" "32 4 128" "128 1 32" "128 1 64" "128 1 128" "128 4 64" "128 4 128"; do
  set -- $cfg
  echo "# B=$1 beam=$2 decode_cus=$3" >> gpurun_out/r05_c17_gen.txt
  timeout 600 python bench.py --generate --batch $1 --beam $2 --decode-cus $3 2>&1 | grep -v amdgpu.ids | tail -2 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); print(json.dumps({k: j[k] for k in ('value', 'serial_value', 'ms_per_step')}), j['roofline']['avg_step_us'], j['roofline']['frac'])
    else:
        print(ln.rstrip()[-300:])
" >> gpurun_out/r05_c17_gen.txt
done
