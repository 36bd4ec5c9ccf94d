#!/bin/bash
# device-scope atomics and L2 requests of the fused SSSP's relax launches (rocprofv3 --pmc, tools/sssp_bench.py)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmcs; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_REQ_sum TCC_MISS_sum --output-format csv -d $O/p -- python3 $R/tools/sssp_bench.py --scale 22 --runs 1 --check 0 > $O/run.log 2>&1
echo "rc=$?"
cd $R
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
f = glob.glob(O + "/p/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_sssp_relax" in r["Kernel_Name"]]
by = collections.defaultdict(dict)
for r in rows: by[int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
top = sorted(by.items(), key=lambda kv: -kv[1].get("TCC_REQ_sum", 0))[:8]
for d, c in sorted(top): print(d, {k: "%.3g" % v for k, v in c.items()})
PY
rm -rf $O/p
