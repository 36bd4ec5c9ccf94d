#!/bin/bash
# Regenerates tests/golden/reference_goldens.json from the reference ITSELF (build container only: needs /root/reference).
# The reference's loader and CPU validators are compiled with plain g++ from the headers where they lie; the three
# moderngpu includes they pull in are served by the host stand-ins under tools/golden_ref/moderngpu/ (own code, for this
# recipe only -- SURVEY appendix A).  Only OUTPUTS are committed: tests/golden/reference_goldens.json (+ the small .mtx
# fixtures that already sit beside it; the larger inputs are re-made deterministically by tests/golden_inputs.py and
# pinned by the sha256 of their text).  This is golden provenance, NOT an oracle/_ref build: nothing here is linked,
# called or shipped by the product, the oracle or the GPU tests.
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
REF=${REF:-/root/reference/gunrock/src}
[ -d "$REF" ] || { echo "no reference tree at $REF: nothing regenerated" >&2; exit 1; }
WORK=$(mktemp -d /tmp/mgx_goldens.XXXXXX)
trap 'rm -rf "$WORK"' EXIT
g++ -std=c++11 -O2 -w -D__device__= -D__host__= -D__forceinline__=inline \
    -I"$ROOT/tools/golden_ref" -I"$REF" "$ROOT/tools/golden_ref/driver.cpp" -o "$WORK/driver"
make -s -C "$ROOT/oracle" liboracle.so
python3 "$ROOT/tools/regen_goldens.py" "$WORK/driver" "$WORK" "$ROOT/tests/golden/reference_goldens.json"
