#!/bin/bash
# mgx_dbfs2_run with G ranks on one GPU over the loopback communicator: wall time per rank and traversal, then the same under
# rocprofv3 --kernel-trace --stats (per-kernel totals per rank and traversal)
# usage: gpu_d2_loopback.sh <scale> <ranks> <exchange> <sources> "<ENV=val ...>" ...   (one pair of runs per argument; "" = defaults)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/d2loop; rm -rf $O; mkdir -p $O
SC=$1; G=$2; EX=$3; NS=$4; shift 4
cd /tmp && export TMPDIR=/tmp
i=0
for cfg in "$@"; do
  i=$((i+1))
  for kv in $cfg; do export "$kv"; done
  echo "=== [$cfg] scale $SC ranks $G exchange $EX" | tee -a $O/summary.txt
  DIST2_CHECK=1 timeout 600 python3 $R/tools/dist2_loopback.py $SC $G $EX $NS > $O/plain$i.log 2>&1; echo "plain rc=$?" >> $O/summary.txt
  grep "^src\|^median\|^check" $O/plain$i.log >> $O/summary.txt
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tr$i -- python3 $R/tools/dist2_loopback.py $SC $G $EX $NS > $O/prof$i.log 2>&1; echo "prof rc=$?" >> $O/summary.txt
  for kv in $cfg; do unset "${kv%%=*}"; done
  f=$(find $O/tr$i -name "*kernel_stats.csv" | head -1)
  python3 - "$f" $G $NS >> $O/summary.txt <<'PY'
import csv, sys
G, NS = int(sys.argv[2]), int(sys.argv[3]) + 1
rows = list(csv.DictReader(open(sys.argv[1])))
keep = ("k_bfs_push_level", "k_d2_", "k_bfs_build", "copyBuffer", "fillBuffer")
tot = 0.0
for r in rows:
    name = r["Name"]
    if not any(k in name for k in keep) or "k_d2_row_facts" in name or "k_d2_owner" in name:
        continue
    calls, ns = int(r["Calls"]), int(r["TotalDurationNs"])
    per = ns / 1e3 / (NS * G)
    tot += per if ("k_bfs" in name or "k_d2_" in name) else 0.0
    print("  %-60s calls %5d  avg %8.1f us  per rank-traversal %8.1f us" % (name[:60], calls, ns / 1e3 / calls, per))
print("  engine kernels per rank and traversal: %.1f us" % tot)
PY
  rm -rf $O/tr$i
done
cat $O/summary.txt
