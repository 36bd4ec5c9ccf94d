#!/bin/bash
# round 6: the neighbour-reduce's layout kernels for ascending SUBSET frontiers (PR's iterations 2, 3): parity, then PR timing with and without
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_nr; rm -rf $O; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -k "neighbour_reduce or pr_matches" > $O/pytest_nr.txt 2>&1; tail -5 $O/pytest_nr.txt
for sw in 1 0; do
  echo "=== MGX_NR_SUBSET=$sw"; MGX_NR_SUBSET=$sw timeout 600 python tools/pr_bench.py 2>&1 | grep -v amdgpu.ids | grep "library layout"
done > $O/pr_s22.txt; cat $O/pr_s22.txt
timeout 300 python bench.py --mode pr --no-cpu-baseline > $O/bench_pr.json 2> $O/bench_pr.err; cut -c1-400 $O/bench_pr.json
