#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
# flags = (level<<8): diag sums only for that level; bit0=0 keeps claims on
MGX_BFS_DIAG=1 MGX_BFS_FLAGS=768 timeout 600 python tools/bfs_levels.py --scale 22 --runs 1 > gpurun_out/levels_diag_hot_l3.log 2>&1
tail -5 gpurun_out/levels_diag_hot_l3.log
MGX_BFS_DIAG=1 MGX_BFS_FLAGS=512 timeout 600 python tools/bfs_levels.py --scale 22 --runs 1 > gpurun_out/levels_diag_hot_l2.log 2>&1
tail -3 gpurun_out/levels_diag_hot_l2.log
