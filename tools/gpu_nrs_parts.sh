#!/bin/bash
# kernel times of the neighbour-reduce's parts (MGX_NR_PARTS: 1 short rows only, 2 long rows only), sliced and unit blocks -> gpurun_out/nrs/parts.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/nrs; mkdir -p $O; rm -f $O/parts.txt
cd /tmp && export TMPDIR=/tmp
for v in ${SLICED:-1 0}; do for p in ${PARTS:-1 2 3}; do
  MGX_NR_SLICED=$v MGX_NR_PARTS=$p timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/tp_${v}_$p -- python3 $R/bench.py --mode pr --scale ${SCALE:-22} --steps 12 --warmup 2 --no-cpu-baseline --no-check > $O/tp_${v}_$p.log 2>&1
  python3 - $O/tp_${v}_$p $v $p >> $O/parts.txt <<'PY'
import csv, glob, sys
d, v, p = sys.argv[1:4]
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_nr" in r["Name"] and "counts" not in r["Name"] and "fill" not in r["Name"]:
            print("sliced=%s parts=%s %-46s calls %4s avg %8.1f us" % (v, p, r["Name"].split("<")[0][-40:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
  rm -rf $O/tp_${v}_$p
done; done
cat $O/parts.txt
