#!/bin/bash
cd $GRAFT_REPO_ROOT
for e in 16 8 216 208; do
echo "== stream ept $e"
MGX_BFS_STREAM_EPT=$e timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-210
MGX_BFS_STREAM_EPT=$e timeout 600 python tools/bfs_levels.py --scale 22 --runs 1 2>&1 | grep -E "slots"
done
