#!/bin/bash
set +e
timeout 3400 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/r04_gputest_b.txt; cat gpurun_out/r04_gputest_b.txt
timeout 300 python tools/probes/adam_keep.py 2>&1 | tail -3
