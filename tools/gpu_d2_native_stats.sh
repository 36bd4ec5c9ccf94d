#!/bin/bash
# per-kernel totals of a partition's rank engines driven in turn by the C++ loop (tools/dist2_single.py <scale> <ranks> native)
# usage: gpu_d2_native_stats.sh <scale> <ranks> "<ENV=val ...>" ...      (one profiled run per argument; "" = defaults)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/d2native; rm -rf $O; mkdir -p $O
SC=$1; G=$2; shift 2
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1))
  for kv in $cfg; do export "$kv"; done
  echo "=== [$cfg] scale $SC ranks $G" | tee -a $O/summary.txt
  timeout 600 python3 $R/tools/dist2_single.py $SC $G native 2>&1 | grep "^native" >> $O/summary.txt
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr$i -- python3 $R/tools/dist2_single.py $SC $G native > $O/run$i.log 2>&1
  echo "profiled rc=$?" >> $O/summary.txt
  for kv in $cfg; do unset "${kv%%=*}"; done
  f=$(find $O/tr$i -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $G >> $O/summary.txt <<'PY'
import csv, sys
G = int(sys.argv[2])
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ("k_bfs_push_level", "k_d2_", "k_bfs_build", "copyBuffer", "fillBuffer")
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
init = [i for i, r in enumerate(rows) if "k_d2_init" in r["Kernel_Name"]]
last = rows[init[-G]:]
t0 = int(last[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in last)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
print("  last traversal: %d launches, span %.1f us, busy %.1f us (gaps %.1f us) -> per rank span %.1f us" % (len(last), (t1 - t0) / 1e3, busy / 1e3, (t1 - t0 - busy) / 1e3, (t1 - t0) / 1e3 / G))
for key in ("k_bfs_push_level", "k_d2_cold_reduce", "k_d2_newbits", "k_d2_lists_apply", "k_d2_or", "k_bfs_build2", "copyBuffer", "fillBuffer"):
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in last if key in r["Kernel_Name"]]
    if d:
        step = G if "Buffer" not in key else len(d)
        print("  last traversal, %-18s n=%3d sum %7.1f us  per level (mean over ranks): %s" % (key, len(d), sum(d), " ".join("%6.1f" % (sum(d[i:i + step]) / len(d[i:i + step])) for i in range(0, len(d), step))))
PY
  rm -rf $O/tr$i
done
cat $O/summary.txt
