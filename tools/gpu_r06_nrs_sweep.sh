#!/bin/bash
# the neighbour-reduce by graph size and number of hot slices (MGX_NR_SLICES caps what the library would take) -> gpurun_out/r06_nrs/sweep.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_nrs; mkdir -p $O; rm -f $O/sweep.txt
cd $R
for sc in ${SCALES:-22 23 24 25}; do for k in ${SLICES:-16 24 32 48 64}; do
  MGX_NR_SLICES=$k timeout 400 python3 bench.py --mode pr --scale $sc --steps 16 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' | tail -1 | python3 -c "
import sys, json
j = json.loads(sys.stdin.read()); s = j.get('nr_slices') or {}
print('RMAT-$sc slices<=$k: %.4f ms  %.1f GTEPS  frac %.3f  parity %s  slices %s rows %s mini-units %s tail %s' % (j['ms_per_step'], j['value']/1e3, j['roofline']['frac'], j.get('parity_vs_oracle'), s.get('hot_slices'), s.get('long_rows'), s.get('mini_units'), s.get('tail_mini_units')))" >> $O/sweep.txt 2>&1
done; done
cat $O/sweep.txt
