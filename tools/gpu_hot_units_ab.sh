#!/bin/bash
# the fused BFS with and without the unit blocks of the hot entries (MGX_BFS_HOT_UNITS), by graph size
for s in ${SCALES:-22 23 24 25}; do for hu in 0 1; do
  steps=64; [ $s -ge 24 ] && steps=16
  MGX_BFS_HOT_UNITS=$hu timeout 500 python bench.py --scale $s --steps $steps --warmup 2 --no-cpu-baseline > gpurun_out/hu_${s}_$hu.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/hu_${s}_$hu.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("scale $s hot_units $hu  ms %.4f  GTEPS %.1f  parity %s  push %.1f  build %.1f  long %.1f  short %.1f  frac %.3f / %.3f" % (d["ms_per_step"], d["value"]/1e3, d["parity_vs_oracle"], r["push_us_per_traversal"], r["build_us_per_traversal"], r["parts"]["long_rows"]["us_per_traversal"], r["parts"]["short_rows"]["us_per_traversal"], r["frac"], r["whole_bfs_frac"]))
PY
done; done
