#!/bin/bash
ulimit -c 0
O=gpurun_out/r2k; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 900 > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -8 $O/pytest_gpu.log
timeout 900 python tools/bfs_ab.py --scale 22 --rounds 2 --configs ";MGX_BFS_VSHORT=0;MGX_BFS_VSHORT=4;MGX_BFS_VSHORT=32" > $O/ab.log 2>&1
echo "ab rc=$?"; cat $O/ab.log
timeout 300 python tools/bfs_levels.py --scale 22 --runs 2 2>&1 | grep -E "^src|level  [2345]|slots|wave" 
