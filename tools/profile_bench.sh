#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_bench.sh <name> [bench args...]
# rocprofv3 --kernel-trace --stats of bench.py -> gpurun_out/<name>_bench_kernel_stats.txt (+ the bench JSON line)
name=$1; shift
args=${*:---steps 9 --warmup 1}
root=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$name
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- python $root/bench.py $args > /tmp/prof_$name.log 2>&1
cd $root
grep '^{"metric"' /tmp/prof_$name.log > gpurun_out/${name}_bench.json
db=$(find /tmp/prof_$name -name '*.db' | head -1)
python tools/rocprof_summary.py "$db" gpurun_out/${name}_bench_kernel_stats.txt "python bench.py $args (+ one-time init), 1x MI355X, under rocprofv3"
