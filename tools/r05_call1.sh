#!/bin/bash
# round 5, GPU call 1: suite + streaming probe + baseline numbers + BertAdam variants (same box)
set +e
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r05_c1_gputest.txt
hipcc --offload-arch=gfx950 -O3 tools/probes/stream_probe.hip -o /tmp/stream_probe 2>/dev/null && /tmp/stream_probe > gpurun_out/r05_stream_probe.txt 2>&1
for v in 0 1 2 3 4; do
  echo "TELL_ADAM_VAR=$v" >> gpurun_out/r05_c1_adam.txt
  TELL_ADAM_VAR=$v python tools/decoder_profile.py faces_objects 32 30 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05_c1_adam.txt
done
python tools/resnet_profile.py 32 20 train 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_c1_resnet.txt
python tools/bench_roberta_gemms.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r05_c1_roberta_gemms.txt
python bench.py --no-cpu-baseline 2> gpurun_out/r05_c1_bench.err | tail -1 > gpurun_out/r05_c1_bench.json
