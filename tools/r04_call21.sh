#!/bin/bash
set +e
timeout 900 python -m pytest tests/test_gpu_encoders.py -q -x -k "stem or resnet" 2>&1 | grep -v "RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" | tail -12
for v in 0 1 0 1; do echo "TELL_STEM_IMPLICIT=$v"; TELL_STEM_IMPLICIT=$v timeout 300 python tools/resnet_profile.py 32 20 train 2>&1 | tail -1; done
TELL_STEM_IMPLICIT=1 timeout 300 python tools/resnet_profile.py 32 20 eval 2>&1 | tail -1
