#!/bin/bash
# final evidence of round 6 in one call: GPU suite, then every profile of tools/r06_profiles.sh
set +e
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed" | tail -2 > gpurun_out/r06_gputest.txt
bash tools/r06_profiles.sh
