#!/bin/bash
# Build libtell_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
# One object per source, compiled in parallel and only when the source (or a header) is newer.
set -e
cd "$(dirname "$0")"
OUT=${1:-libtell_hip.so}
SRCS="api gemm gemm_pp2 gemm_q4 gemm_q4e gemm_s64 elementwise layernorm dynconv attention adaptive optim conv encoders lstm multi decode"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value"
mkdir -p _obj
pids=""
for s in $SRCS; do
  if [ ! -f _obj/$s.o ] || [ $s.hip -nt _obj/$s.o ] || [ common.h -nt _obj/$s.o ] || [ gemm_common.h -nt _obj/$s.o ] || [ gemm_epi.h -nt _obj/$s.o -a "${s#gemm}" != "$s" ] || [ gemm_q4_loop.inc -nt _obj/$s.o -a "${s#gemm_q4}" != "$s" ] || [ gemm_q4e_loop.inc -nt _obj/$s.o -a "$s" = gemm_q4e ] || [ ../../include/tell_hip.h -nt _obj/$s.o ]; then
    hipcc $FLAGS -c $s.hip -o _obj/$s.o &
    pids="$pids $!"
  fi
done
for p in $pids; do wait $p; done
objs=""
for s in $SRCS; do objs="$objs _obj/$s.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$OUT"
echo "built $(pwd)/$OUT"
