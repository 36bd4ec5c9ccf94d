#!/bin/bash
# per-kernel totals of the rank engines of a partitioned traversal sharing one GPU (tools/dist2_single.py) under several settings
# usage: gpu_d2_stats.sh <scale> <ranks> "<ENV=val ...>" ...      (one profiled run per argument; "" = defaults)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/d2stats; rm -rf $O; mkdir -p $O
SC=$1; G=$2; shift 2
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1))
  for kv in $cfg; do export "$kv"; done
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr$i -- python3 $R/tools/dist2_single.py $SC $G > $O/run$i.log 2>&1
  echo "=== [$cfg] rc=$?" | tee -a $O/summary.txt
  for kv in $cfg; do unset "${kv%%=*}"; done
  grep "^lists\|^gather" $O/run$i.log | tail -5 >> $O/summary.txt
  f=$(find $O/tr$i -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $G >> $O/summary.txt <<'PY'
import csv, sys
G = int(sys.argv[2])
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ("k_bfs_push_level", "k_d2_", "k_bfs_build", "copyBuffer", "fillBuffer", "CatArrayBatchedCopy", "elementwise")
tot = 0.0
for r in rows:
    name = r["Name"]
    if not any(k in name for k in keep) or "k_d2_row_facts" in name or "k_d2_owner" in name:
        continue
    calls, ns = int(r["Calls"]), int(r["TotalDurationNs"])
    per = ns / 1e3 / (6 * G)          # six traversals, G rank engines
    tot += per if ("k_bfs" in name or "k_d2_" in name) else 0.0
    print("  %-60s calls %5d  avg %8.1f us  per rank-traversal %8.1f us" % (name[:60], calls, ns / 1e3 / calls, per))
print("  engine kernels per rank and traversal: %.1f us" % tot)
PY
  t=$(find $O/tr$i -name "*kernel_trace.csv" | head -1)
  python3 - "$t" $G >> $O/summary.txt <<'PY'
import csv, sys
G = int(sys.argv[2])
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last traversal: every engine kernel by level (G calls per level: the rank engines one after the other)
init = [i for i, r in enumerate(rows) if "k_d2_init" in r["Kernel_Name"]]
last = rows[init[-G]:]
for key in ("k_bfs_push_level", "k_d2_newbits", "k_d2_lists_apply", "k_d2_or", "k_bfs_build2", "k_d2_merge_build"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last if key in r["Kernel_Name"]]
    if d:
        print("  last traversal, %-18s per level (mean over ranks): %s" % (key, " ".join("%6.1f" % (sum(d[i:i + G]) / len(d[i:i + G])) for i in range(0, len(d), G))))
nb = len([r for r in last if "k_d2_newbits" in r["Kernel_Name"]])
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last if "k_bfs_push_level" in r["Kernel_Name"]]
if nb and len(d) == 3 * nb:      # MGX_DIST_PUSH_SPLIT=1: three launches per rank and level
    for part, name in enumerate(("cold pass", "long rows", "short rows")):
        dd = d[part::3]
        print("  split push, %-10s per level (mean over ranks): %s" % (name, " ".join("%6.1f" % (sum(dd[i:i + G]) / len(dd[i:i + G])) for i in range(0, len(dd), G))))
PY
  rm -rf $O/tr$i
done
cat $O/summary.txt
