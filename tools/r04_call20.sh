#!/bin/bash
set +e
timeout 3400 python -m pytest tests -m gpu -q 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -6 > gpurun_out/r04_gputest_b.txt; cat gpurun_out/r04_gputest_b.txt
SEQ_ANCHOR=bertadam bash tools/profile_cmd.sh r04_decoder "decoder half of the step alone (fwd + loss + bwd + BertAdam), B=32, step graph: python tools/decoder_profile.py faces_objects 32 20" python tools/decoder_profile.py faces_objects 32 20
bash tools/profile_bench.sh r04 --steps 9 --warmup 1 --no-cpu-baseline --no-secondary --no-generation --no-loader
python tools/bench_roberta_gemms.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_roberta_gemms.txt
python tools/bench_attention.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r04_attention.txt
python bench.py 2> gpurun_out/r04_bench.err | tail -1 > gpurun_out/r04_bench.json
python -c "
import json; d=json.load(open('gpurun_out/r04_bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('isolated'), d['decoder_step']['alone_ms'], d['generation']['beam4']['value'], d['generation']['greedy']['value'], d['loader_variable_lengths']['value'], d['loader_variable_lengths']['first_epochs_value'], d['secondary']['value'])"
cat gpurun_out/r04_roberta_gemms.txt gpurun_out/r04_attention.txt gpurun_out/r04_decoder_stdout.txt
