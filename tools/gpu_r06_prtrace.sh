#!/bin/bash
# kernel sequence of the LAST PR enact of tools/pr_bench.py (library layout): where the 3 iterations' time goes
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_prtrace; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/tools/pr_bench.py > $O/log.txt 2>&1
cd $R
python3 - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06_prtrace"
f = glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
hits = [i for i, r in enumerate(rows) if 'k_nr_values_subset' in r['Kernel_Name']]
i0 = hits[-2] - 12          # the last enact: iteration 0 (full) is a few kernels in front of the first subset call
prev = None; t0 = int(rows[i0]['Start_Timestamp'])
with open(O + "/pr_sequence.txt", "w") as out:
    for r in rows[i0:]:
        st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        name = r['Kernel_Name'].replace('void mgx::', '').replace('mgx::', '').replace('gunrock::', '')[:90]
        line = "%9.1f us  %-90s dur %8.1f us  gap %7.1f us" % ((st - t0) / 1e3, name, (en - st) / 1e3, (st - prev) / 1e3 if prev else 0)
        print(line); out.write(line + "\n")
        prev = en
PY
