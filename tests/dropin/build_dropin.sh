#!/bin/bash
# Drop-in check of the operator boundary: compile the REFERENCE's own, unmodified
#   gunrock/tests/{bfs,sssp,pr,kcore}/test_*.cu   (drivers, with their CPU validation calls)
#   gunrock/src/{bfs,sssp,pr,kcore}/*_{problem,functor,enactor}.hxx
# (kcore is outside the hot-path scope, SURVEY 8f.4: it is here because it is free extra coverage of the operator
#  API -- has_output=false advance with an atomicAdd functor, filter with three different functors -- with the
#  reference's own CPU oracle inside the driver)
# against THIS repo's data model + operator headers (include/gunrock/{graph,frontier,problem,
# enactor,intrinsics,advance,filter,neighborhood,test_utils}.hxx, kernels in include/mgx/).
# Nothing of the reference is copied: its files are read where they lie under /root/reference
# through a symlink farm in a temp dir; only the binaries land in tests/dropin/_bin/
# (git-ignored, shipped to the GPU box like any other built artefact).
# Only runs where /root/reference exists (this container); the GPU box uses the prebuilt files.
set -e
REF=${REF:-/root/reference}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$ROOT/tests/dropin/_bin
[ -d "$REF/gunrock/src" ] || { echo "no reference tree at $REF: skipping"; exit 0; }
FARM=$(mktemp -d)
trap 'rm -rf "$FARM"' EXIT
mkdir -p "$FARM/inc" "$OUT"
ln -s "$ROOT/include/mgx" "$FARM/mgx"
ln -s "$ROOT/include/mgx.h" "$FARM/mgx.h"
for f in graph frontier problem enactor intrinsics advance filter neighborhood test_utils; do
  ln -s "$ROOT/include/gunrock/$f.hxx" "$FARM/inc/$f.hxx"
done
ln -s "$ROOT/include/gunrock/moderngpu" "$FARM/inc/moderngpu"
for d in bfs sssp pr kcore; do ln -s "$REF/gunrock/src/$d" "$FARM/inc/$d"; done
for t in bfs sssp pr kcore; do
  ${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O2 -std=c++17 -Wno-unused-value -x hip \
      -I"$FARM/inc" "$REF/gunrock/tests/$t/test_$t.cu" -o "$OUT/ref_test_$t"
  echo "built $OUT/ref_test_$t"
done
# the repo's own timing driver around the reference's UNCHANGED BFS enactor / problem / functor (tools/dropin_cost.py)
${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O2 -std=c++17 -Wno-unused-value -x hip \
    -I"$FARM/inc" "$ROOT/tests/dropin/ref_bench_bfs.cu" -o "$OUT/ref_bench_bfs"
echo "built $OUT/ref_bench_bfs"
