#!/bin/bash
# generation pipeline (generate_stream) A/B: pipelined vs serial, batch 32 / 64 / 128, greedy and beam 4
set +e
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_data_eval.py -x -q 2>&1 | tail -5 > gpurun_out/r05_c16_tests.txt
: > gpurun_out/r05_c16_gen.txt
for B in 32 64 128; do
  for beam in 1 4; do
    echo "# B=$B beam=$beam" >> gpurun_out/r05_c16_gen.txt
    timeout 600 python bench.py --generate --batch $B --beam $beam 2>&1 | grep -v amdgpu.ids | tail -3 >> gpurun_out/r05_c16_gen.txt
  done
done
