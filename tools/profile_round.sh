#!/bin/bash
# Produces the evidence bundle of a round under gpurun_out/round/: test log, smoke, bench lines (BFS batch / per call, SSSP, PR,
# direction-optimising sweep), rocprofv3 kernel-trace stats of the SAME bench commands, FETCH/WRITE PMC passes for the three
# dominant kernels, per-level / per-iteration logs, a timeline of single traversals.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round
rm -rf $O; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -q -m gpu --timeout 900 > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
MGX_LIB=$R/mini_amd/libmgx_lab.so timeout 900 python -m pytest tests -q -m gpu --timeout 600 -k "variants or cold_edge_pass_vs or lds_distance" > $O/pytest_gpu_lab.log 2>&1
echo "pytest (lab library) rc=$?"; tail -2 $O/pytest_gpu_lab.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1
echo "smoke rc=$?"; tail -1 $O/smoke.log
# the PMC passes FIRST: the traffic figures they give (tied to the hash of the sources) are what the bench lines of this
# very run report as roofline.traffic
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-check > $O/pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/pmc_sssp_$c -- python3 $R/bench.py --mode sssp --steps 4 --warmup 1 --no-cpu-baseline --no-check > $O/pmc_sssp_$c.log 2>&1
  echo "pmc sssp $c rc=$?"
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/pmc_pr_$c -- python3 $R/bench.py --mode pr --steps 4 --warmup 1 --no-cpu-baseline --no-check > $O/pmc_pr_$c.log 2>&1
  echo "pmc pr $c rc=$?"
done
cd $R
python3 tools/summarize_profiles.py $O > /dev/null 2>&1
[ -s $O/pmc_traffic.json ] && cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
timeout 900 python bench.py > $O/bench.log 2>&1
echo "bench rc=$?"; tail -1 $O/bench.log | cut -c1-300
timeout 900 python bench.py --per-call --steps 64 > $O/bench_per_call.log 2>&1
timeout 900 python bench.py --mode sssp > $O/bench_sssp.log 2>&1
echo "bench sssp rc=$?"; tail -1 $O/bench_sssp.log | cut -c1-300
timeout 900 python bench.py --mode pr > $O/bench_pr.log 2>&1
echo "bench pr rc=$?"; tail -1 $O/bench_pr.log | cut -c1-300
timeout 600 python tools/bfs_levels.py --scale 22 --runs 2 > $O/levels.log 2>&1
timeout 600 python tools/sssp_iterations.py --runs 2 > $O/sssp_iterations.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-check > $O/trace_bench.log 2>&1
echo "trace rc=$?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_sssp -- python3 $R/bench.py --mode sssp --steps 8 --warmup 2 --no-cpu-baseline --no-check > $O/trace_sssp.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_pr -- python3 $R/bench.py --mode pr --steps 16 --warmup 2 --no-cpu-baseline --no-check > $O/trace_pr.log 2>&1
timeout 900 rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/pmc_TCC -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-check > $O/pmc_TCC.log 2>&1
cd $R
for m in sssp pr; do
  f=$(ls $O/trace_$m/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python3 - "$f" > $O/kernel_stats_$m.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
w = csv.DictWriter(sys.stdout, fieldnames=rows[0].keys()); w.writeheader()
for r in rows:
    if "mgx" in r["Name"] or float(r["Percentage"]) > 1.0:
        r = dict(r); r["Name"] = r["Name"][:120]; w.writerow(r)
PY
done
bash tools/gpu_timeline.sh "" > $O/timeline.txt 2>&1
# the partitioned engine on this one GPU: G rank engines in turn, no exchange time
# (lists: the level protocol of mgx_dbfs2_run, the default; gather: bitmaps on every level -- what rounds 2 and 3 measured)
for cfg in "25 8" "26 8" "22 8" "22 1"; do
  set -- $cfg
  timeout 900 python tools/dist2_single.py $1 $2 2>/dev/null | grep -v amdgpu.ids | tail -6 > $O/dist2_single_$1_$2.log
done
# (round 5) the same engines driven in turn by the C++ loop itself: a whole traversal enqueued ahead from the level plan, one host wait
for cfg in "22 1" "22 2" "22 4" "22 8" "25 8" "26 8"; do
  set -- $cfg
  timeout 900 python tools/dist2_single.py $1 $2 native 2>/dev/null | grep -v amdgpu.ids | tail -6 > $O/dist2_native_$1_$2.log
done
bash tools/gpu_d2_native_stats.sh 26 8 "" > /dev/null 2>&1
cp $R/gpurun_out/d2native/summary.txt $O/dist2_native_kernels_26_8.txt 2>/dev/null
bash tools/gpu_d2_native_stats.sh 22 8 "" > /dev/null 2>&1
cp $R/gpurun_out/d2native/summary.txt $O/dist2_native_kernels_22_8.txt 2>/dev/null
timeout 600 python tools/dist2_loopback.py 22 8 reduce 8 2>/dev/null | grep -v amdgpu.ids | tail -10 > $O/dist2_loopback_22_8.log
# (round 5) RMAT-26 on one GPU, the graphs off RMAT-22, what the drop-in costs, the timed batch per kernel
timeout 900 python bench.py --scale 26 --steps 8 --warmup 1 > $O/bench_rmat26_1gpu.log 2>&1
for g in "uniform 22 16" "grid2d 22 4" "rmat 23 32" "rmat 24 16" "rmat 25 16" "rmat 20 32"; do
  set -- $g
  timeout 600 python bench.py --graph $1 --scale $2 --steps $3 --warmup 2 --cpu-seconds 5 > $O/bench_$1_$2.log 2>&1
done
# (round 6) the neighbour-reduce above RMAT-22: hot slices by graph size
for sc in 23 24 25; do
  timeout 600 python bench.py --mode pr --scale $sc --steps 16 --warmup 2 --cpu-seconds 5 > $O/bench_pr_rmat_$sc.log 2>&1
done
timeout 600 python tools/dropin_cost.py 20 4 2>&1 | grep -v amdgpu.ids > $O/dropin_cost.log
timeout 600 python tools/dropin_cost.py 22 4 2>&1 | grep -v amdgpu.ids >> $O/dropin_cost.log
bash tools/gpu_trace_batch.sh "" 64 > /dev/null 2>&1
cp $R/gpurun_out/trb/batch_stats.txt $O/batch_stats.txt 2>/dev/null
python3 tools/pmc_by_kernel.py $O 5 > $O/pmc_by_kernel.txt 2>&1
timeout 900 python tools/dist2_single.py 26 8 gather 2>/dev/null | grep -v amdgpu.ids | tail -6 > $O/dist2_single_26_8_gather.log
timeout 900 python tools/dist2_single.py 26 8 reduce 2>/dev/null | grep -v amdgpu.ids | tail -6 > $O/dist2_single_26_8_reduce.log
# the rank engines' kernels per rank and traversal, per level, and the three parts of the push grid
bash tools/gpu_d2_stats.sh 26 8 "" "MGX_DIST_PUSH_SPLIT=1" > /dev/null 2>&1
cp $R/gpurun_out/d2stats/summary.txt $O/dist2_kernels_26_8.txt 2>/dev/null
DIST2_CHECK=1 timeout 900 python tools/dist2_single.py 23 4 2>/dev/null | grep -v amdgpu.ids | tail -8 > $O/dist2_single_23_4_check.log
# ... and its kernels under rocprofv3
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_dist2 -- python3 $R/tools/dist2_single.py 25 8 > $O/trace_dist2.log 2>&1
cd $R
f=$(ls $O/trace_dist2/*/*kernel_stats.csv 2>/dev/null | head -1)
[ -n "$f" ] && python3 - "$f" > $O/kernel_stats_dist2_25_8.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
w = csv.DictWriter(sys.stdout, fieldnames=rows[0].keys()); w.writeheader()
for r in rows:
    if "mgx" in r["Name"]:
        r = dict(r); r["Name"] = r["Name"][:120]; w.writerow(r)
PY
# the driver's own command line
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2>&1
# auxiliary logs: operator path (reference loop, idempotent mode), direction-optimising sweep
timeout 600 python tools/bfs_operator_bench.py 22 > $O/bfs_operator_s22.log 2>&1
timeout 600 python tools/sssp_bench.py > $O/sssp_s22.log 2>&1
timeout 600 python tools/pr_bench.py > $O/pr_s22.log 2>&1
timeout 600 python tools/kcore_bench.py 20 > $O/kcore_s20.log 2>&1
for a in 0.01 1 4 16 64 256 1000; do
  timeout 300 python bench.py --mode do --alpha $a --steps 16 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('alpha=$a MTEPS %.2f ms_per_step %.4f (one call per source: %.4f) parity_vs_oracle %s' % (j['value'], j['ms_per_step'], j['per_call']['ms_per_step'], j.get('parity_vs_oracle')))" >> $O/dobfs_alpha_sweep.txt
done
rm -rf $O/trace*/*/*kernel_trace.csv
python3 tools/summarize_profiles.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
# what goes under profiles/ (the caller copies gpurun_out/round/keep/* to profiles/rNN/ and pmc_traffic.json to profiles/)
mkdir -p $O/keep
cp $O/summary.txt $O/summary.json $O/kernel_stats_mgx.csv $O/kernel_stats_sssp.csv $O/kernel_stats_pr.csv $O/levels.log $O/sssp_iterations.log $O/timeline.txt $O/keep/ 2>/dev/null
cp $O/pmc_traffic.json $O/bfs_operator_s22.log $O/dobfs_alpha_sweep.txt $O/sssp_s22.log $O/pr_s22.log $O/kcore_s20.log $O/keep/ 2>/dev/null
cp $O/dist2_single_*.log $O/kernel_stats_dist2_25_8.csv $O/dist2_kernels_26_8.txt $O/keep/ 2>/dev/null
cp $O/dist2_native_*.log $O/dist2_native_kernels_*.txt $O/dist2_loopback_22_8.log $O/dropin_cost.log $O/batch_stats.txt $O/pmc_by_kernel.txt $O/keep/ 2>/dev/null
grep '^{' $O/bench_rmat26_1gpu.log | tail -1 > $O/keep/bench_line_rmat26_1gpu.json
for g in uniform_22 grid2d_22 rmat_23 rmat_24 rmat_25 rmat_20; do grep '^{' $O/bench_$g.log | tail -1 > $O/keep/bench_line_$g.json; done
grep '^{' $O/bench_driver_cmd.log | tail -1 > $O/keep/bench_line_driver_cmd.json
grep '^{' $O/bench.log | tail -1 > $O/keep/bench_line.json
grep '^{' $O/bench_per_call.log | tail -1 > $O/keep/bench_line_per_call.json
grep '^{' $O/bench_sssp.log | tail -1 > $O/keep/bench_line_sssp.json
grep '^{' $O/bench_pr.log | tail -1 > $O/keep/bench_line_pr.json
for sc in 23 24 25; do grep '^{' $O/bench_pr_rmat_$sc.log | tail -1 > $O/keep/bench_line_pr_rmat_$sc.json; done
(echo "product library (mini_amd/libmgx.so), python -m pytest tests -q -m gpu:"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1
 echo "lab library (MGX_LIB=mini_amd/libmgx_lab.so), the variant tests:"; grep -E "passed|failed" $O/pytest_gpu_lab.log | tail -1) > $O/keep/pytest_gpu_tail.txt
