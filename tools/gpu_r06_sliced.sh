#!/bin/bash
# round 6: hub levels by slice of the destinations (bfs_fused_sliced.hpp): parity, per-level times with and without, the bench both ways
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_sliced; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "hub_levels_by_slice" > $O/pytest_sliced.txt 2>&1; tail -15 $O/pytest_sliced.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bfs and not large" > $O/pytest_bfs.txt 2>&1; tail -3 $O/pytest_bfs.txt
timeout 600 python tools/bfs_levels_plain.py --scale 22 --sources 4 --configs "MGX_BFS_SLICED=0;MGX_BFS_SLICED=1;MGX_BFS_SLICED=1,MGX_BFS_SLICED_MIN_EDGES=8000000" 2>&1 | grep -v amdgpu.ids > $O/levels.txt
cat $O/levels.txt
for i in 1 2 3; do
  for sw in 0 1; do
    MGX_BFS_SLICED=$sw timeout 300 python bench.py --no-cpu-baseline > $O/bench_$sw_$i.json 2> $O/err.txt || tail -3 $O/err.txt
    python - $O/bench_$sw_$i.json $sw <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j.get("roofline",{})
print("sliced=%s ms %.4f per_call %.4f push %.1f build %.1f frac %.3f slot_frac %.3f bytes %d parity %s" % (sys.argv[2], j["ms_per_step"], j.get("per_call",{}).get("ms_per_step",0), r.get("push_us_per_traversal",0), r.get("build_us_per_traversal",0), r.get("frac",0), r.get("slot_frac",0), j["layout"]["device_bytes"], j.get("parity_vs_oracle")))
PY
  done
done
