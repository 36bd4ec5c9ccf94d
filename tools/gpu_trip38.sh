#!/bin/bash
cd $GRAFT_REPO_ROOT
for fl in 0 784 770 769; do
echo "== flags $fl (level 3: 16=no test, 2=no ds_or, 1=no stores)"
MGX_BFS_FLAGS=$fl timeout 600 python tools/bfs_levels.py --scale 22 --runs 1 2>&1 | grep -E "slots"
done
