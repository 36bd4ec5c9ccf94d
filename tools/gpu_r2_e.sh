#!/bin/bash
ulimit -c 0
O=gpurun_out/r2e; rm -rf $O; mkdir -p $O
for cfg in "" "MGX_BFS_BIGLDS=1" "MGX_BFS_BIGLDS=1 MGX_BFS_DENSE_DIAG=1" "MGX_BFS_DENSE_DIAG=1"; do
  echo "=== $cfg" >> $O/levels.log
  env $cfg timeout 300 python tools/bfs_levels.py --scale 22 --runs 2 2>&1 | grep -E "^src|level  [234]|slots|stream" >> $O/levels.log
done
cat $O/levels.log
