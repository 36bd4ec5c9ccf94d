#!/bin/bash
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r2j; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -- python3 $R/tools/bfs_ab.py --scale 22 --rounds 1 --steps 6 --warmup 2 --mode 1 --alpha 64 --configs "" > $O/run.log 2>&1
tail -3 $O/run.log
python3 $R/tools/trace_window.py $O/t k_bfs_fused_init 4 30
