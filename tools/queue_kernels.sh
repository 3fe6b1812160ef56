root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_qk
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_qk -o qk -- python $root/bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-roofline > /tmp/qk.log 2>&1
grep '"metric"' /tmp/qk.log | cut -c1-160
cd $root
python tools/queue_kernels.py /tmp/prof_qk 10 $1
