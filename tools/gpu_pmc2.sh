#!/bin/bash
# per-dispatch counters of the fused BFS kernels: bash tools/gpu_pmc2.sh "<counter list 1>" "<counter list 2>" ...
ulimit -c 0
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc2; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/tools/bfs_ab.py --scale 22 --rounds 1 --steps 2 --warmup 1 --configs "" > $O/run$i.log 2>&1
  echo "pmc pass $i ($set) rc=$?"
done
python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
for d in sorted(glob.glob(O + "/p*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        # group by dispatch id, keep the LAST traversal's kernels
        disp = collections.OrderedDict()
        for r in rows:
            if "k_bfs" not in r["Kernel_Name"]: continue
            k = int(r["Dispatch_Id"])
            disp.setdefault(k, {"name": r["Kernel_Name"]})[r["Counter_Name"]] = float(r["Counter_Value"])
        ids = sorted(disp)
        inits = [i for i in ids if "fused_init" in disp[i]["name"]]
        last = [i for i in ids if i >= inits[-1]]
        for i in last:
            x = disp[i]
            nm = x["name"].replace("void mgx::", "").replace("mgx::", "")[:28]
            print("%-28s " % nm + "  ".join("%s=%.4g" % (k, v) for k, v in x.items() if k != "name"))
PY
