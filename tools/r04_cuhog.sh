#!/bin/bash
# DP contention rehearsal on one GPU: bench.py --cu-hog N, static tile lists vs per-XCD tile counters
set +e
mkdir -p gpurun_out
out=gpurun_out/r04_cu_hog.txt
echo "# python bench.py --cu-hog N --no-cpu-baseline --no-secondary --no-generation --no-loader --no-dp-selftest (configs[2], 20 steps); TELL_Q4_DYNAMIC=0/1" > $out
echo "# N CUs held | q4 tile order | samples/s | ms/step | dominant GEMM in-region us | isolated us" >> $out
for n in 0 8 16 32; do for d in 0 1; do
  TELL_Q4_DYNAMIC=$d timeout 600 python bench.py --cu-hog $n --no-cpu-baseline --no-secondary --no-generation --no-loader --no-dp-selftest 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline',{})
print('%3d | %s | %8.1f | %6.3f | %6.1f | %6.1f' % ($n, 'per-XCD counters' if $d else 'static lists     ', d['value'], d['ms_per_step'], r.get('avg_launch_us') or -1, (r.get('isolated') or {}).get('avg_launch_us') or -1))" >> $out
done; done
cat $out
