#!/bin/bash
# LDS / issue counters of k_nrs_edges with only the long rows' part (MGX_NR_PARTS=2) -> gpurun_out/r06_nr_lds/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_nr_lds; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU SQ_INSTS_SALU" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  MGX_NR_PARTS=${PARTS:-2} timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/p_$tag -- python3 $R/bench.py --mode pr --steps 4 --warmup 1 --no-cpu-baseline --no-check > $O/p_$tag.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/r06_nr_lds"
acc = {}
for f in glob.glob(O + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_nrs_edges" in k or "k_nrs_fold" in k:
            key = (k.split("<")[0].split("::")[-1], r["Counter_Name"])
            a = acc.setdefault(key, [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print("%-14s %-30s per dispatch %16.0f  (%d dispatches)" % (k, c, v / n, n))
PY
