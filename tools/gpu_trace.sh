#!/bin/bash
# kernel-trace stats of the bench command -> gpurun_out/trace_quick/kernel_stats.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_quick; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-check "$@" > $O/bench.log 2>&1
echo "trace rc=$?"; tail -1 $O/bench.log | cut -c1-300
cd $R
python3 - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/trace_quick"
for f in glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(O + "/kernel_stats.txt", "w") as g:
        for r in rows:
            if "mgx" in r["Name"] or float(r["Percentage"]) > 0.5:
                line = "%-90s calls %6s avg %9.1f us total %9.3f ms  %5s%%" % (r["Name"][:90], r["Calls"], float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e6, r["Percentage"])
                print(line); g.write(line + "\n")
PY
