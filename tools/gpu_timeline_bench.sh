#!/bin/bash
# timeline of single traversals, any bench.py arguments: gpu_timeline_bench.sh "<bench args>" [occurrences...]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tl2; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-check $1 > $O/run.log 2>&1
echo "trace rc=$?"
shift
for occ in "$@"; do
  echo "--- traversal occurrence $occ"; python3 $R/tools/trace_window.py $O/trace k_bfs_fused_init $occ 24
done > $O/windows.txt 2>&1
rm -rf $O/trace
cat $O/windows.txt
