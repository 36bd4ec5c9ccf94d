#!/bin/bash
cd $GRAFT_REPO_ROOT
ulimit -c 0
for sh in 4 5 0; do
echo "== stream shape $sh"
MGX_BFS_STREAM_SHAPE=$sh timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 600 -x -k "bfs" 2>&1 | tail -1
MGX_BFS_STREAM_SHAPE=$sh timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-210
MGX_BFS_STREAM_SHAPE=$sh timeout 600 python tools/bfs_levels.py --scale 22 --runs 2 2>&1 | grep -E "slots"
done
