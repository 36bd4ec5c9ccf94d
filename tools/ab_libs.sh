#!/bin/bash
# A/B of two builds of the library on one box: bench.py (RMAT-22 push) alternating between MGX_LIB=$1 and MGX_LIB=$2,
# $3 rounds (default 2); prints ms per traversal, per-call, push / build us and the long / short parts of every run.
# usage (GPU box): bash tools/ab_libs.sh mini_amd/libmgx_base.so mini_amd/libmgx.so [rounds] [extra bench args]
A=$1; B=$2; R=${3:-2}; shift 3 2>/dev/null
O=gpurun_out/ab_libs; mkdir -p $O
for i in $(seq 1 $R); do
  for L in $A $B; do
    tag=$(basename $L .so)_$i
    MGX_LIB=$L timeout 300 python bench.py --no-cpu-baseline "$@" > $O/$tag.json 2> $O/$tag.err || echo "$tag failed: $(tail -n 2 $O/$tag.err)"
  done
done
python - "$O" <<'PY'
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        j = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    r = j.get("roofline", {})
    p = r.get("parts", {})
    print("%-22s ms %.4f  per_call %.4f  push %.1f  build %.1f  long %.1f  short %.1f  slot_frac %.3f  frac %.3f  parity %s" % (
        os.path.basename(f)[:-5], j["ms_per_step"], j.get("per_call", {}).get("ms_per_step", 0), r.get("push_us_per_traversal", 0),
        r.get("build_us_per_traversal", 0), p.get("long_rows", {}).get("us_per_traversal", 0), p.get("short_rows", {}).get("us_per_traversal", 0),
        r.get("slot_frac", 0), r.get("frac", 0), j.get("parity_vs_oracle")))
PY
