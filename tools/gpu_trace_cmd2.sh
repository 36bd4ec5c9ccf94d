#!/bin/bash
# kernel-trace stats of an arbitrary python tool: bash tools/gpu_trace_cmd2.sh <outdir-name> <script> [args]
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; shift; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
S=$R/$1; shift
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $S "$@" > $O/run.log 2>&1
echo "trace rc=$?"; tail -6 $O/run.log
cd $R
python3 - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
for f in glob.glob(O + "/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open(O + "/kernel_stats.txt", "w") as g:
        for r in rows:
            if float(r["Percentage"]) > 0.3:
                line = "%-100s calls %6s avg %9.1f us total %9.3f ms  %5s%%" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e6, r["Percentage"])
                print(line); g.write(line + "\n")
PY
