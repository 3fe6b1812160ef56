#!/bin/bash
# Build libtell_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -e
cd "$(dirname "$0")"
OUT=${1:-libtell_hip.so}
SRCS="api.hip gemm.hip elementwise.hip layernorm.hip dynconv.hip attention.hip adaptive.hip optim.hip conv.hip encoders.hip"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-result $SRCS -o "$OUT"
echo "built $(pwd)/$OUT"
