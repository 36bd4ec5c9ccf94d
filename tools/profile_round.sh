#!/bin/bash
# Produces the evidence bundle of a round under gpurun_out/round/: test log, smoke, bench line,
# rocprofv3 kernel-trace stats of the SAME bench command, and FETCH/WRITE PMC passes.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/round
rm -rf $O; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests -q -m gpu --timeout 900 > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1
echo "smoke rc=$?"; tail -1 $O/smoke.log
timeout 300 ./tools/microbench > $O/microbench.jsonl 2>&1
timeout 300 ./tools/microbench4 > $O/microbench4.jsonl 2>&1
# the PMC passes FIRST: the traffic figure they give (tied to the hash of the sources) is what the bench line of this
# very run reports as roofline.traffic
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-check > $O/pmc_$c.log 2>&1
  echo "pmc $c rc=$?"
done
cd $R
python3 tools/summarize_profiles.py $O > /dev/null 2>&1
[ -s $O/pmc_traffic.json ] && cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
timeout 900 python bench.py --steps 16 --warmup 2 > $O/bench.log 2>&1
echo "bench rc=$?"; tail -1 $O/bench.log
timeout 600 python tools/bfs_levels.py --scale 22 --runs 2 > $O/levels.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-check > $O/trace_bench.log 2>&1
echo "trace rc=$?"
timeout 900 rocprofv3 --pmc TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/pmc_TCC -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-check > $O/pmc_TCC.log 2>&1
cd $R
# auxiliary logs: operator path (reference loop, idempotent mode), SSSP (operator / fused / near-far), direction-optimising sweep
timeout 600 python tools/bfs_operator_bench.py 22 > $O/bfs_operator_s22.log 2>&1
timeout 600 python tools/sssp_bench.py --scale 22 --runs 4 --check 0 2>&1 | grep -E "^SSSP" > $O/sssp_s22.log
for a in 0.01 1 4 16 64 256 1000; do
  timeout 300 python bench.py --mode do --alpha $a --steps 16 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('alpha=$a MTEPS %.2f ms_per_step %.4f parity_vs_oracle %s' % (j['value'], j['ms_per_step'], j.get('parity_vs_oracle')))" >> $O/dobfs_alpha_sweep.txt
done
rm -rf $O/trace/*/*kernel_trace.csv
python3 tools/summarize_profiles.py $O > $O/summary.txt 2>&1
cat $O/summary.txt
# what goes under profiles/ (the caller copies gpurun_out/round/keep/* to profiles/rNN/ and pmc_traffic.json to profiles/)
mkdir -p $O/keep
cp $O/summary.txt $O/summary.json $O/kernel_stats_mgx.csv $O/levels.log $O/microbench.jsonl $O/microbench4.jsonl $O/keep/ 2>/dev/null
cp $O/pmc_traffic.json $O/bfs_operator_s22.log $O/sssp_s22.log $O/dobfs_alpha_sweep.txt $O/keep/ 2>/dev/null
grep '^{' $O/bench.log | tail -1 > $O/keep/bench_line.json
