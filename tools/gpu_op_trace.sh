#!/bin/bash
# kernels of the operator-per-superstep BFS (tools/bfs_operator_bench.py) under rocprofv3: totals per kernel
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/optrace; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr -- python3 $R/tools/bfs_operator_bench.py 22 > $O/run.log 2>&1
grep "operator path\|fused path" $O/run.log
f=$(find $O/tr -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:22]:
    print("%-110s calls %5d avg %9.1f us total %9.1f us  %5.1f%%" % (r["Name"][:110], int(r["Calls"]), float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e3, float(r["Percentage"])))
PY
rm -rf $O/tr
