#!/bin/bash
# usage (GPU box, repo root): tools/profile_cmd.sh <name> <note> <command...>
# rocprofv3 --kernel-trace --stats of an arbitrary command -> gpurun_out/<name>_kernel_stats.txt (+ the command's stdout)
name=$1; note=$2; shift; shift
root=$(pwd)
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$name
( cd $root && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o $name -- "$@" > /tmp/prof_$name.log 2>&1 )
cd $root
grep -v "^W2026\|^I2026\|^E2026" /tmp/prof_$name.log | tail -5 > gpurun_out/${name}_stdout.txt
db=$(find /tmp/prof_$name -name '*.db' | head -1)
python tools/rocprof_summary.py "$db" gpurun_out/${name}_kernel_stats.txt "$note"
if [ -n "$SEQ_ANCHOR" ]; then python tools/rocprof_sequence.py "$db" gpurun_out/${name}_sequence.txt "$SEQ_ANCHOR"; fi
