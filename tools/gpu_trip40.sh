#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_sssp; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/sssp_bench.py > $O/log.txt 2>&1
tail -3 $O/log.txt
cd $R
python3 - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/trace_sssp"
for f in glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.7:
            print("%-100s calls %6s avg %9.1f us total %9.3f ms %s%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
