#!/bin/bash
# Build libgnpde_hip.so (gfx950 only) next to the sources.  Called by __graft_entry__.build().
set -e
cd "$(dirname "$0")"
ROOT="$(cd ../.. && pwd)"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I${ROOT}/include -I. -Wall -Wno-unused-function -Wno-pass-failed"
mkdir -p build
objs=""
pids=""
for f in error.cpp graph_prep.cpp spmm.hip linear.hip attention.hip fused_attn.hip backward.hip adjoint.hip solver.hip misc.hip early_stop.hip sharded.hip rewire.hip twohop.hip dopri5.hip adjoint_adaptive.hip graph_device.hip; do
  o="build/${f%.*}.o"
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ common.h -nt "$o" ] || [ epilogue.h -nt "$o" ] || [ rhs.h -nt "$o" ] || [ "${ROOT}/include/gnpde.h" -nt "$o" ]; then
    rm -f "$o"
    $HIPCC $FLAGS -c "$f" -o "$o" &
    pids="$pids $!"
  fi
  objs="$objs $o"
done
fail=0
for p in $pids; do wait "$p" || fail=1; done
[ "$fail" = 0 ] || { echo "compilation failed" >&2; exit 1; }
for o in $objs; do [ -f "$o" ] || { echo "missing $o" >&2; exit 1; }; done
$HIPCC --offload-arch=gfx950 -shared -fPIC $objs -o libgnpde_hip.so
echo "built $(pwd)/libgnpde_hip.so"
