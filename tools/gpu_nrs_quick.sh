#!/bin/bash
# the sliced neighbour-reduce: parity tests, kernel times of the parts, one bench line -> gpurun_out/nrs/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/nrs; mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "neighbour_reduce or pr_matches" > $O/pytest_nr.log 2>&1; tail -2 $O/pytest_nr.log
SLICED="${SLICED:-1 0}" PARTS="${PARTS:-3 1}" bash tools/gpu_nrs_parts.sh
timeout 300 python3 bench.py --mode pr --steps 32 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 | python3 -c "
import sys, json
j = json.loads(sys.stdin.read()); print('bench pr: %.4f ms  frac %.3f  parity %s' % (j['ms_per_step'], j['roofline']['frac'], j.get('parity_vs_oracle')))"
