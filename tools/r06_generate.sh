#!/bin/bash
# Round-6 generation evidence in one GPU-box call (repo root): bash tools/r06_generate.sh <tag> -> gpurun_out/<tag>_*
tag=${1:-r06}
set +e
mkdir -p gpurun_out
for b in 1 4; do python bench.py --generate --beam $b 2>/dev/null | tail -1 > gpurun_out/${tag}_generate_beam$b.json; done
SEQ_ANCHOR=greedy_update bash tools/profile_cmd.sh ${tag}_generate "greedy generation, B=32: python bench.py --generate --beam 1 --steps 1 --warmup 1" python bench.py --generate --beam 1 --steps 1 --warmup 1
SEQ_ANCHOR=beam_update bash tools/profile_cmd.sh ${tag}_beam "beam-4 generation, B=32: python bench.py --generate --beam 4 --steps 1 --warmup 1" python bench.py --generate --beam 4 --steps 1 --warmup 1
bash tools/pmc_generate_traffic.sh 1 gpurun_out/${tag}_pmc_generate_greedy_traffic.json > /dev/null 2>&1
bash tools/pmc_generate_traffic.sh 4 gpurun_out/${tag}_pmc_generate_beam4_traffic.json > /dev/null 2>&1
python tools/bench_skinny.py 32 2>&1 | grep -v amdgpu.ids > gpurun_out/${tag}_skinny_bench.txt
python tools/bench_skinny.py 128 2>&1 | grep -v amdgpu.ids >> gpurun_out/${tag}_skinny_bench.txt
