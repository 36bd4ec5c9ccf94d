#!/bin/bash
# round 2, first GPU call: the whole GPU suite on the slot scheme, then A/B of the chain / unit-block switches
ulimit -c 0
O=gpurun_out/r2a; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 900 > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -15 $O/pytest_gpu.log
timeout 600 python tools/bfs_ab.py --scale 22 --rounds 2 > $O/ab.log 2>&1
echo "ab rc=$?"; cat $O/ab.log
timeout 300 python tools/bfs_levels.py --scale 22 --runs 2 > $O/levels.log 2>&1
echo "levels rc=$?"; cat $O/levels.log
