#!/bin/bash
# Late bundle of round 5 on the sources with the sliced neighbour-reduce: the PMC passes and bench lines of tools/profile_refresh.sh
# (-> gpurun_out/round2/keep), the whole GPU suite (product + lab library), smoke, kernel-trace stats of the three bench commands,
# a fuzz campaign -> gpurun_out/final/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
cd $R
ulimit -c 0
GRAPHS="rmat 23 32;rmat 24 16;rmat 20 32" bash tools/profile_refresh.sh > $O/refresh.log 2>&1; tail -3 $O/refresh.log
timeout 1500 python -m pytest tests -q -m gpu --timeout 900 > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -2 $O/pytest_gpu.log
MGX_LIB=$R/mini_amd/libmgx_lab.so timeout 600 python -m pytest tests -q -m gpu --timeout 600 -k "variants or cold_edge_pass_vs or lds_distance" > $O/pytest_gpu_lab.log 2>&1
echo "pytest (lab library) rc=$?"; tail -1 $O/pytest_gpu_lab.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1
echo "smoke rc=$?"; tail -1 $O/smoke.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-check > $O/trace_bench.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_sssp -- python3 $R/bench.py --mode sssp --steps 8 --warmup 2 --no-cpu-baseline --no-check > $O/trace_sssp.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_pr -- python3 $R/bench.py --mode pr --steps 16 --warmup 2 --no-cpu-baseline --no-check > $O/trace_pr.log 2>&1
cd $R
for m in "" _sssp _pr; do
  f=$(ls $O/trace$m/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" > $O/kernel_stats${m:-_mgx}.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
w = csv.DictWriter(sys.stdout, fieldnames=rows[0].keys()); w.writeheader()
for r in rows:
    if "mgx" in r["Name"] or float(r["Percentage"]) > 1.0:
        r = dict(r); r["Name"] = r["Name"][:120]; w.writerow(r)
PY
done
rm -rf $O/trace $O/trace_sssp $O/trace_pr
FUZZ_SEED=50607 timeout 400 python3 tools/fuzz_parity.py 150 > $O/fuzz_seed50607.log 2>&1; tail -2 $O/fuzz_seed50607.log
ls $O
