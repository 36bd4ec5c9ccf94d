#!/bin/bash
cd $GRAFT_REPO_ROOT
ulimit -c 0
for w in 3 1 3 1; do
echo "== wave shape $w"
MGX_BFS_WAVE_SHAPE=$w timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-210
done
