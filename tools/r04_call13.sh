#!/bin/bash
set +e
timeout 200 python tools/probes/q4_residual.py "res q4" 2>&1 | tail -8
timeout 900 python -m pytest tests/test_gpu_step_graph.py -q -x 2>&1 | tail -5
