#!/bin/bash
# timeline of single traversals on the product path: rocprofv3 kernel trace of a short bench run, windows printed
# usage: gpu_timeline.sh "<ENV=val ENV=val>" ...   (one run per argument; "" = defaults)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tl; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1))
  for kv in $cfg; do export "$kv"; done
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace$i -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-check > $O/run$i.log 2>&1
  echo "=== [$cfg] trace rc=$?" | tee -a $O/windows.txt
  for kv in $cfg; do unset "${kv%%=*}"; done
  for occ in 4 7; do
    echo "--- traversal occurrence $occ"; python3 $R/tools/trace_window.py $O/trace$i k_bfs_fused_init $occ 15
  done >> $O/windows.txt 2>&1
  rm -rf $O/trace$i
done
cat $O/windows.txt
