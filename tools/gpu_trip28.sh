#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 600 -x 2>&1 | tail -2
for fl in 0 8; do
echo "== flags $fl"
MGX_BFS_FLAGS=$fl timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-check 2>&1 | tail -1 | cut -c1-210
MGX_BFS_FLAGS=$fl timeout 600 python tools/bfs_levels.py --scale 22 --runs 1 2>&1 | grep -E "level  [0-9]|slots"
done
