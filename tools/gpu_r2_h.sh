#!/bin/bash
ulimit -c 0
O=gpurun_out/r2h; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 900 > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest_gpu.log
timeout 900 python tools/bfs_ab.py --scale 22 --rounds 2 --configs ";MGX_BFS_BUILD_LIST=1" > $O/ab.log 2>&1
echo "ab rc=$?"; cat $O/ab.log
bash tools/gpu_trace.sh > /dev/null 2>&1; python3 tools/trace_one.py gpurun_out/trace_quick 4
