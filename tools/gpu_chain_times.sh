#!/bin/bash
# durations of the in-place chain launches per traversal (tools/chain_debug.py under a rocprofv3 kernel trace)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/cd; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tools/chain_debug.py 22 > $O/run.log 2>&1
cd $R
python3 - "$O" <<'PY'
import csv, glob, sys
O = sys.argv[1]
f = glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
seq = []
for r in rows:
    n = r['Kernel_Name']
    if 'k_bfs_fused_init' in n: seq.append([])
    if seq and ('chain_inplace' in n): seq[-1].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print("chain_inplace durations per traversal (us):")
for s in seq: print("  ", " ".join("%.1f" % x for x in s))
PY
rm -rf $O/trace
grep "^src" $O/run.log | cut -c1-200
