#!/bin/bash
set +e
timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r04_gputest_b.txt; cat gpurun_out/r04_gputest_b.txt
python tools/bench_roberta_gemms.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_roberta_gemms.txt; cat gpurun_out/r04_roberta_gemms.txt
python tools/bench_attention.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_attention.txt; cat gpurun_out/r04_attention.txt
python bench.py 2> gpurun_out/r04_bench.err | tail -1 > gpurun_out/r04_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r04_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('isolated'), d['roofline'].get('traffic'), d['decoder_step']['alone_ms'], d['generation']['beam4']['value'], d['generation']['greedy']['value'], d['loader_variable_lengths']['value'], d['loader_variable_lengths']['first_epochs_value'])"
