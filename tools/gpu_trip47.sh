#!/bin/bash
cd $GRAFT_REPO_ROOT
ulimit -c 0
for f in 64 0 64 0; do
echo "== flags $f"
MGX_BFS_FLAGS=$f timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-210
done
