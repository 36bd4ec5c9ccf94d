#!/bin/bash
ulimit -c 0
O=gpurun_out/r2c; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 900 > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
timeout 900 python tools/bfs_ab.py --scale 22 --rounds 2 --configs ";MGX_BFS_CHAIN_MAX_EDGES=2048;MGX_BFS_CHAIN_MAX_EDGES=512;MGX_BFS_CHAIN_MAX_EDGES=0;MGX_BFS_DENSE=4;MGX_BFS_DENSE=64;MGX_BFS_DENSE=1000000" > $O/ab.log 2>&1
echo "ab rc=$?"; cat $O/ab.log
