#!/bin/bash
# Build libtell_hip.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
# One object per source, compiled in parallel and only when the source (or a header) is newer.
#   build.sh                -> libtell_hip.so (the shipped library)
#   PROBES=1 build.sh       -> libtell_hip_probes.so: the same sources with -DTELL_PROBES, i.e. plus the wrong-result timing
#                              ablations that tools/probes/ switch on through tell_set_option (csrc/options.h); never loaded
#                              by the package unless TELL_LIB points at it
set -e
cd "$(dirname "$0")"
if [ -n "$PROBES" ]; then OBJ=_obj_probes; DEF="-DTELL_PROBES"; OUT=${1:-libtell_hip_probes.so}; else OBJ=_obj; DEF=""; OUT=${1:-libtell_hip.so}; fi
SRCS="api gemm gemm_pp2 gemm_q4 gemm_q4e gemm_s64 elementwise layernorm dynconv attention adaptive optim conv encoders lstm multi decode"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -Wno-unused-value $DEF"
mkdir -p $OBJ
pids=""
for s in $SRCS; do
  o=$OBJ/$s.o
  if [ ! -f $o ] || [ $s.hip -nt $o ] || [ common.h -nt $o ] || [ options.h -nt $o ] || [ gemm_common.h -nt $o ] || [ gemm_epi.h -nt $o -a "${s#gemm}" != "$s" ] || [ gemm_q4_loop.inc -nt $o -a "${s#gemm_q4}" != "$s" ] || [ gemm_q4e_loop.inc -nt $o -a "$s" = gemm_q4e ] || [ ../../include/tell_hip.h -nt $o ]; then
    hipcc $FLAGS -c $s.hip -o $o &
    pids="$pids $!"
  fi
done
for p in $pids; do wait $p; done
objs=""
for s in $SRCS; do objs="$objs $OBJ/$s.o"; done
hipcc --offload-arch=gfx950 -shared -fPIC $objs -o "$OUT"
echo "built $(pwd)/$OUT"
