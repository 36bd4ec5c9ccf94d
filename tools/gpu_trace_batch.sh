#!/bin/bash
# kernel trace of the default bench command's timed batch, summarised per traversal (tools/trace_batch_stats.py)
# usage: gpu_trace_batch.sh "<extra bench args>" [steps]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trb; rm -rf $O; mkdir -p $O
ST=${2:-64}
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps $ST --warmup 2 --no-cpu-baseline --no-check $1 > $O/run.log 2>&1
echo "trace rc=$?"
python3 $R/tools/trace_batch_stats.py $O/trace $ST > $O/batch_stats.txt 2>&1
rm -rf $O/trace
cat $O/batch_stats.txt
