#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
MGX_BFS_DIAG=1 MGX_BFS_FLAGS=769 timeout 600 python tools/bfs_levels.py --scale 22 --runs 1 > gpurun_out/levels_diag_noclaim.log 2>&1
echo "diag rc=$?"; tail -13 gpurun_out/levels_diag_noclaim.log
