#!/bin/bash
# round 5, the sliced neighbour-reduce: its parity tests, the A/B against the unit blocks (MGX_NR_SLICED=0), kernel times, and the
# BFS headline on the same sources (the epilogue's pinned loads) -> gpurun_out/nrs/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/nrs; rm -rf $O; mkdir -p $O
cd $R
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "neighbour_reduce or pr_matches" > $O/pytest_nr.log 2>&1; tail -3 $O/pytest_nr.log
for v in 1 0 1 0; do
  MGX_NR_SLICED=$v timeout 300 python3 bench.py --mode pr --steps 32 --warmup 2 --no-cpu-baseline > $O/pr_$v.log 2>&1
  grep '^{' $O/pr_$v.log | tail -1 | python3 -c "
import sys, json
j = json.loads(sys.stdin.read()); print('MGX_NR_SLICED=$v: %.4f ms  frac %.3f  parity %s  slices %s build %.2fs' % (j['ms_per_step'], j['roofline']['frac'], j.get('parity_vs_oracle'), j.get('nr_slices'), j.get('nr_slices_build_s', -1)))" >> $O/ab.log 2>&1 || tail -5 $O/pr_$v.log >> $O/ab.log
done
for i in 1 2; do
  timeout 300 python3 bench.py --steps 64 --warmup 2 --cpu-seconds 3 > $O/push_$i.log 2>&1
  grep '^{' $O/push_$i.log | tail -1 | python3 -c "
import sys, json
j = json.loads(sys.stdin.read()); print('push run $i: %.4f ms  %.1f GTEPS  parity %s  push frac %.3f slot %.3f whole %.3f' % (j['ms_per_step'], j['value']/1e3, j.get('parity_vs_oracle'), j['roofline']['frac'], j['roofline']['slot_frac'], j['roofline']['whole_bfs_frac']))" >> $O/ab.log 2>&1 || tail -5 $O/push_$i.log >> $O/ab.log
done
cat $O/ab.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "kernel_variants" > $O/pytest_variants.log 2>&1; tail -3 $O/pytest_variants.log
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  MGX_NR_SLICED=$v timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_pr_$v -- python3 $R/bench.py --mode pr --steps 16 --warmup 2 --no-cpu-baseline --no-check > $O/trace_pr_$v.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, os
O = os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/nrs"
for v in (1, 0):
    for f in glob.glob(O + "/trace_pr_%d/**/*kernel_stats.csv" % v, recursive=True):
        with open(O + "/kernel_stats_pr_%d.txt" % v, "w") as g:
            for r in csv.DictReader(open(f)):
                if "k_nr" in r["Name"]:
                    line = "sliced=%d %-70s calls %6s avg %9.1f us" % (v, r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3)
                    print(line); g.write(line + "\n")
    os.system("rm -rf %s/trace_pr_%d" % (O, v))
PY
