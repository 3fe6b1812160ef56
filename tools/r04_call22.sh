#!/bin/bash
set +e
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "attention or attn" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -4
python tools/bench_attention.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_attention.txt | tail -5
timeout 300 python tools/decoder_profile.py faces_objects 32 20 2>&1 | tail -2 | head -1
timeout 1500 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_train.py tests/test_gpu_fullsize.py -q -x 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -3
