#!/bin/bash
# parity subset, then tools/bfs_ab.py in direction-optimising mode at alpha 16 / 64 / 256 over the switch settings in $1
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2do; rm -rf $O; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_streams.py -q -x -m gpu --timeout 600 > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
for al in 16 64 256; do
timeout 600 python tools/bfs_ab.py --scale 22 --rounds 3 --steps 16 --warmup 2 --mode 1 --alpha $al --configs "$1" > $O/ab$al.log 2>&1
echo "alpha $al"; grep best $O/ab$al.log
done
