#!/bin/bash
# the fused SSSP's sliced heavy iterations: parity tests, then the bench line with and without (MGX_SSSP_SLICED2) -> gpurun_out/ssl/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ssl; rm -rf $O; mkdir -p $O
cd $R
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "sssp" > $O/pytest_sssp.log 2>&1; tail -4 $O/pytest_sssp.log
for v in ${VALS:-1 0 1}; do
  MGX_SSSP_SLICED2=$v timeout 400 python3 bench.py --mode sssp --steps 16 --warmup 2 --cpu-seconds 3 > $O/sssp_$v.log 2>&1
  grep '^{' $O/sssp_$v.log | tail -1 | python3 -c "
import sys, json
j = json.loads(sys.stdin.read()); print('MGX_SSSP_SLICED2=$v: %.4f ms  frac %.3f  parity %s' % (j['ms_per_step'], j['roofline']['frac'], j.get('parity_vs_oracle')))" >> $O/ab.log 2>&1 || tail -5 $O/sssp_$v.log >> $O/ab.log
done
cat $O/ab.log
MGX_SSSP_SLICED2=1 timeout 300 python3 tools/sssp_iterations.py > $O/iters_1.log 2>&1; grep -v amdgpu.ids $O/iters_1.log | head -12
