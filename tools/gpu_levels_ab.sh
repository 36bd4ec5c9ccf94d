#!/bin/bash
# tools/bfs_levels.py (per-level / per-part timing) with and without a switch
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/lv; rm -rf $O; mkdir -p $O; cd $R
for cfg in "" "$1"; do
  for kv in $cfg; do export "$kv"; done
  echo "=== [$cfg]"; timeout 300 python tools/bfs_levels.py --scale 22 --runs 2 2>&1 | grep -v amdgpu.ids
  for kv in $cfg; do unset "${kv%%=*}"; done
done > $O/levels.txt 2>&1
cat $O/levels.txt
