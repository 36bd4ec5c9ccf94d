#!/bin/bash
cd $GRAFT_REPO_ROOT
for fl in 513 514 516 520; do
echo "== flags $fl"
MGX_BFS_FLAGS=$fl timeout 600 python tools/bfs_levels.py --scale 22 --runs 2 2>&1 | grep -E "level  [2]" 
done
