root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_tl
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_tl -o tl -- python $root/bench.py --steps 12 --warmup 6 --no-cpu-baseline --no-roofline > /tmp/tl.log 2>&1
tail -1 /tmp/tl.log | cut -c1-200
cd $root
python tools/timeline.py /tmp/prof_tl 120
