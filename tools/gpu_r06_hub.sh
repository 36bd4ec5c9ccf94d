#!/bin/bash
# round 6, experiment 2: what the hub level's flush + OR cost (lab: deferred marks dropped, OR skipped), and a non-returning claim in the queue walk
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_hub; rm -rf $O; mkdir -p $O; cd $R
MGX_LIB=$R/mini_amd/libmgx_lab.so timeout 600 python tools/bfs_levels_plain.py --scale 22 --sources 4 --configs ";MGX_BFS_DENSE_DIAG=8;MGX_BFS_DENSE_DIAG=8,MGX_BFS_BUILD_DIAG=32;MGX_BFS_DEFER=0" 2>&1 | grep -v amdgpu.ids > $O/levels_lab.txt
cat $O/levels_lab.txt
MGX_LIB=$R/mini_amd/libmgx_noret.so timeout 600 python tools/bfs_levels_plain.py --scale 22 --sources 4 --configs ";" 2>&1 | grep -v amdgpu.ids > $O/levels_noret.txt
cat $O/levels_noret.txt
bash tools/ab_libs.sh mini_amd/libmgx.so mini_amd/libmgx_noret.so 2 > $O/ab_noret.txt 2>&1; cat $O/ab_noret.txt
