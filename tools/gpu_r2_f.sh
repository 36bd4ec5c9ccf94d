#!/bin/bash
ulimit -c 0
O=gpurun_out/r2f; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 900 > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
timeout 900 python tools/bfs_ab.py --scale 22 --rounds 2 --configs ";MGX_BFS_DEFER=0;MGX_BFS_DEFER=512;MGX_BFS_DEFER=8192;MGX_BFS_DEFER=1" > $O/ab.log 2>&1
echo "ab rc=$?"; cat $O/ab.log
for cfg in "" "MGX_BFS_DEFER=0"; do
  echo "=== $cfg" >> $O/levels.log
  env $cfg timeout 300 python tools/bfs_levels.py --scale 22 --runs 2 2>&1 | grep -E "^src|level  [234]|slots|stream" >> $O/levels.log
done
cat $O/levels.log
