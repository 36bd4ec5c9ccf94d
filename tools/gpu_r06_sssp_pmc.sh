#!/bin/bash
# per-dispatch counters of k_sssp_relax_dense (the sweep) and k_sssp_relax: atomics, L2 requests / misses, stalls -> gpurun_out/r06_sssp_pmc/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_sssp_pmc; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "TCC_EA0_ATOMIC_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_WRREQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU" "TCC_ATOMIC_sum TCC_READ_sum TCC_WRITE_sum TCP_TCC_ATOMIC_WITH_RET_REQ_sum"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/p_$tag -- python3 $R/tools/sssp_iterations.py --runs 1 > $O/p_$tag.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06_sssp_pmc"
rows = {}
for f in glob.glob(O + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_sssp_relax" in k:
            name = "dense" if "relax_dense" in k else "walk"
            rows.setdefault((name, r["Counter_Name"]), []).append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
names = sorted({c for (_, c) in rows})
for kind in ("walk", "dense"):
    print("== k_sssp_relax%s: the LAST traversal's dispatches in order (iteration 0 ..)" % ("_dense" if kind == "dense" else ""))
    cols = {}
    for c in names:
        v = sorted(rows.get((kind, c), []))
        cols[c] = [x for _, x in v][-12:]
    n = max(len(v) for v in cols.values()) if cols else 0
    print("it  " + "  ".join("%22s" % c[:22] for c in names))
    for i in range(n):
        print("%2d  " % i + "  ".join("%22.0f" % (cols[c][i] if i < len(cols[c]) else -1) for c in names))
PY
