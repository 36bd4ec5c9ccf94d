#!/bin/bash
# the fused SSSP's sliced heavy iterations: parity tests, then the bench line by share of the edges an iteration must hold
# (MGX_SSSP_SLICED2_SHARE; MGX_SSSP_SLICED2=0: the unit blocks only) -> gpurun_out/ssl/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ssl; rm -rf $O; mkdir -p $O
cd $R
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "sssp" > $O/pytest_sssp.log 2>&1; tail -2 $O/pytest_sssp.log
for v in ${VALS:-off 0.75 0.5 0.9 0.0 off 0.75}; do
  if [ $v = off ]; then export MGX_SSSP_SLICED2=0; else export MGX_SSSP_SLICED2=1 MGX_SSSP_SLICED2_SHARE=$v; fi
  timeout 400 python3 bench.py --mode sssp --steps 16 --warmup 2 --no-cpu-baseline > $O/sssp_$v.log 2>&1
  grep '^{' $O/sssp_$v.log | tail -1 | python3 -c "
import sys, json
j = json.loads(sys.stdin.read()); print('share $v: %.4f ms  frac %.3f  parity %s' % (j['ms_per_step'], j['roofline']['frac'], j.get('parity_vs_oracle')))" >> $O/ab.log 2>&1 || tail -5 $O/sssp_$v.log >> $O/ab.log
done
cat $O/ab.log
export MGX_SSSP_SLICED2=1 MGX_SSSP_SLICED2_SHARE=0.75
timeout 300 python3 tools/sssp_iterations.py > $O/iters_1.log 2>&1; grep -v amdgpu.ids $O/iters_1.log | head -26
