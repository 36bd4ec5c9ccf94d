#!/bin/bash
# where the unit-block body's time goes: marks off / test off
ulimit -c 0
O=gpurun_out/r2b; rm -rf $O; mkdir -p $O
for cfg in "" "MGX_BFS_DENSE=1000000" "MGX_BFS_DENSE=1000000 MGX_BFS_DENSE_DIAG=1" "MGX_BFS_DENSE=1000000 MGX_BFS_DENSE_DIAG=2" "MGX_BFS_DENSE=0"; do
  echo "=== $cfg" >> $O/levels.log
  env $cfg timeout 300 python tools/bfs_levels.py --scale 22 --runs 2 2>&1 | grep -E "^src|level  [1234]|slots|stream" >> $O/levels.log
done
cat $O/levels.log
