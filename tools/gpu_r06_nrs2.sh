#!/bin/bash
# the slices policy by graph size: parity tests of the neighbour-reduce, the default cut at R-MAT 22 .. 25, three PR iterations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_nrs; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "neighbour_reduce or pr_matches or segreduce" > $O/pytest_nr.txt 2>&1; tail -2 $O/pytest_nr.txt
rm -f $O/default.txt
for sc in 20 22 23 24 25; do
  timeout 400 python3 bench.py --mode pr --scale $sc --steps 16 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 | python3 -c "
import sys, json
j = json.loads(sys.stdin.read()); s = j.get('nr_slices') or {}
print('RMAT-$sc: %.4f ms  %.1f GTEPS  frac %.3f  parity %s  slices %s rows %s mini-units %s tail %s' % (j['ms_per_step'], j['value']/1e3, j['roofline']['frac'], j.get('parity_vs_oracle'), s.get('hot_slices'), s.get('long_rows'), s.get('mini_units'), s.get('tail_mini_units')))" >> $O/default.txt 2>&1
done
cat $O/default.txt
