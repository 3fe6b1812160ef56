#!/bin/bash
# configs[2] step timeline under the four schedules (repo root, GPU box) -> gpurun_out/r06_step_timeline.txt
out=gpurun_out/${1:-r06}_step_timeline.txt
: > $out
for rs in own main; do for dyn in 0 1; do
  TELL_RESNET_STREAM=$rs TELL_Q4_DYNAMIC=$dyn python tools/step_timeline.py 2>&1 | grep -v amdgpu.ids >> $out
  echo >> $out
done; done
for rs in own main; do for dyn in 0 1; do
  echo "# bench.py (5 windows of 20 steps), TELL_RESNET_STREAM=$rs TELL_Q4_DYNAMIC=$dyn" >> $out
  TELL_RESNET_STREAM=$rs TELL_Q4_DYNAMIC=$dyn python bench.py --no-cpu-baseline --no-secondary --no-generation --no-loader --no-dp-selftest --no-pmc 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('value %.1f samples/s, %.3f ms/step, windows %s; q4 in-region %.1f us (frac %.3f), alone %.1f us (frac %.3f); decoder alone %.3f ms' % (d['value'], d['ms_per_step'], d['windows_ms_per_step'], r['avg_launch_us'], r['frac'], r['isolated']['avg_launch_us'], r['isolated']['frac'], d['decoder_step']['alone_ms']))" >> $out
done; done
