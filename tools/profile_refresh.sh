R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round2; rm -rf $O; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "long_chain or run_many or per_source or mid_size" > $O/pytest.log 2>&1; tail -2 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-check > $O/pmc_$c.log 2>&1
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/pmc_sssp_$c -- python3 $R/bench.py --mode sssp --steps 4 --warmup 1 --no-cpu-baseline --no-check > $O/pmc_sssp_$c.log 2>&1
  timeout 900 rocprofv3 --pmc $c --output-format csv -d $O/pmc_pr_$c -- python3 $R/bench.py --mode pr --steps 4 --warmup 1 --no-cpu-baseline --no-check > $O/pmc_pr_$c.log 2>&1
done
cd $R
python3 tools/summarize_profiles.py $O > $O/summ.log 2>&1
[ -s $O/pmc_traffic.json ] && cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json
timeout 900 python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-200
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.log 2>&1
timeout 900 python bench.py --per-call --steps 64 > $O/bench_per_call.log 2>&1
timeout 900 python bench.py --mode sssp > $O/bench_sssp.log 2>&1
timeout 900 python bench.py --mode pr > $O/bench_pr.log 2>&1
IFS=';' read -ra GRAPH_LIST <<< "${GRAPHS:-uniform 22 16;grid2d 22 4;rmat 23 32;rmat 24 16;rmat 25 16;rmat 20 32}"
for g in "${GRAPH_LIST[@]}"; do
  set -- $g
  timeout 600 python bench.py --graph $1 --scale $2 --steps $3 --warmup 2 --cpu-seconds 5 > $O/bench_$1_$2.log 2>&1
done
python3 tools/pmc_by_kernel.py $O 5 > $O/pmc_by_kernel.txt 2>&1
mkdir -p $O/keep
cp $O/pmc_traffic.json $O/pmc_by_kernel.txt $O/keep/
grep '^{' $O/bench.log | tail -1 > $O/keep/bench_line.json
grep '^{' $O/bench_driver_cmd.log | tail -1 > $O/keep/bench_line_driver_cmd.json
grep '^{' $O/bench_per_call.log | tail -1 > $O/keep/bench_line_per_call.json
grep '^{' $O/bench_sssp.log | tail -1 > $O/keep/bench_line_sssp.json
grep '^{' $O/bench_pr.log | tail -1 > $O/keep/bench_line_pr.json
for g in uniform_22 grid2d_22 rmat_23 rmat_24 rmat_25 rmat_20; do [ -s $O/bench_$g.log ] && grep '^{' $O/bench_$g.log | tail -1 > $O/keep/bench_line_$g.json; done
rm -rf $O/pmc_*
ls $O/keep
