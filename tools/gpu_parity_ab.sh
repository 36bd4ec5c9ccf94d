#!/bin/bash
# parity subset (tests/test_gpu_parity.py, test_gpu_configs.py), then tools/bfs_ab.py on RMAT-22 over the switch settings in $1
#   gpurun -- bash tools/gpu_parity_ab.sh ";MGX_BFS_COLD=0" [noparity]
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2l; rm -rf $O; mkdir -p $O
cd $R
if [ "$2" != "noparity" ]; then
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -x -m gpu --timeout 600 > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
fi
timeout 600 python tools/bfs_ab.py --scale 22 --rounds 3 --steps 16 --warmup 2 --configs "$1" > $O/ab.log 2>&1
grep best $O/ab.log
