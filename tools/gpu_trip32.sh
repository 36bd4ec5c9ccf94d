#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace_dist; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/dist2_single.py 25 8 > $O/log.txt 2>&1
tail -2 $O/log.txt
cd $R
python3 - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/trace_dist"
for f in glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "mgx" in r["Name"] or "k_d2" in r["Name"]:
            print("%-80s calls %6s avg %9.1f us total %9.3f ms" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e6))
PY
