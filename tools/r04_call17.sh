#!/bin/bash
set +e
timeout 600 python -m pytest tests/test_gpu_ops.py -q -x -k "dynconv_block" 2>&1 | tail -3
for a in 0 1 2 4 6 7; do echo "ABL=$a"; TELL_DCB_ABL=$a timeout 300 python tools/bench_dynconv.py 2>&1 | grep -v amdgpu.ids | tail -1; done
