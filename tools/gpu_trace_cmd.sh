#!/bin/bash
# kernel-trace stats of an arbitrary python tool:  gpu_trace_cmd.sh tools/xyz.py [args]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_cmd; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/"$@" > $O/log.txt 2>&1
tail -4 $O/log.txt
cd $R
python3 - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/trace_cmd"
for f in glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 1.0:
            print("%-110s calls %6s avg %9.1f us total %9.3f ms %s%%" % (r["Name"][:110], r["Calls"], float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
PY
