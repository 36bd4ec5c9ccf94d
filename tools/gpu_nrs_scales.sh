#!/bin/bash
# the neighbour-reduce by graph size, sliced long rows against unit blocks -> gpurun_out/nrs/scales.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/nrs; mkdir -p $O; rm -f $O/scales.txt
cd $R
for sc in ${SCALES:-20 23 24 25}; do for v in 1 0; do
  MGX_NR_SLICED=$v timeout 400 python3 bench.py --mode pr --scale $sc --steps 16 --warmup 2 --cpu-seconds 2 2>&1 | grep '^{' | tail -1 | python3 -c "
import sys, json
j = json.loads(sys.stdin.read()); s = j.get('nr_slices') or {}
print('RMAT-$sc sliced=$v: %.4f ms  %.1f GTEPS  frac %.3f  parity %s  mini-units %s tail %s' % (j['ms_per_step'], j['value']/1e3, j['roofline']['frac'], j.get('parity_vs_oracle'), s.get('mini_units'), s.get('tail_mini_units')))" >> $O/scales.txt 2>&1
done; done
cat $O/scales.txt
FUZZ_SEED=60708 timeout 330 python3 tools/fuzz_parity.py 240 > $O/fuzz_seed60708.log 2>&1; tail -1 $O/fuzz_seed60708.log
