root=$(pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_cn
timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_cn -o cn -- python $root/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-roofline > /tmp/cn.log 2>&1
cd $root
python tools/copy_neighbours.py /tmp/prof_cn
