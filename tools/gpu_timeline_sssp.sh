#!/bin/bash
# kernel sequence of the fused SSSP loop (tools/sssp_bench.py under a rocprofv3 kernel trace)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ss; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tools/sssp_bench.py --scale 22 --runs 2 --check 0 > $O/run.log 2>&1
echo "trace rc=$?"; tail -3 $O/run.log
cd $R
python3 - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
f = glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_sssp' in r['Kernel_Name']]
# last 40 sssp kernels
prev = None
for i in idx[-44:]:
    r = rows[i]; st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("%-60s dur %8.1f us gap %7.1f" % (r['Kernel_Name'].replace('void mgx::','')[:60], (en-st)/1e3, (st-prev)/1e3 if prev else 0)); prev = en
PY
rm -rf $O/trace
