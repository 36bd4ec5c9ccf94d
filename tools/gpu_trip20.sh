#!/bin/bash
cd $GRAFT_REPO_ROOT
for lm in 8 16 32 64; do
echo "== long_min $lm"
MGX_BFS_LONG_MIN=$lm timeout 600 python tools/bfs_levels.py --scale 22 --runs 1 2>&1 | grep -E "level  [2-4]|stream|wave" | head -8
MGX_BFS_LONG_MIN=$lm timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-check 2>&1 | tail -1 | cut -c1-200
done
echo "== ept 8"
MGX_BFS_STREAM_EPT=8 timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-check 2>&1 | tail -1 | cut -c1-200
