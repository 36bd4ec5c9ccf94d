#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2g; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for d in 0 1 4 8 16 31; do
  MGX_BFS_BUILD_DIAG=$d timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t$d -- python3 $R/tools/bfs_ab.py --scale 22 --rounds 1 --steps 4 --warmup 1 --configs "" > $O/run$d.log 2>&1
  echo "== build_diag $d"; python3 $R/tools/trace_window.py $O/t$d k_bfs_fused_init 1 6 | grep -E "build|push"
  rm -rf $O/t$d
done
