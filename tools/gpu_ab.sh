#!/bin/bash
# A/B of an environment switch on one box: bench line per setting.  usage: VAR=MGX_X VALS="0 1 0 1" bash tools/gpu_ab.sh
ulimit -c 0
mkdir -p gpurun_out; rm -f gpurun_out/ab.log
for v in $VALS; do
  export $VAR=$v
  timeout 300 python3 bench.py --steps 16 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import sys, json
l = sys.stdin.read().strip()
try:
    j = json.loads(l); print('$VAR=$v: %.4f ms  %.1f GTEPS  parity %s  roofline frac %.3f' % (j['ms_per_step'], j['value']/1e3, j.get('parity_vs_oracle'), j['roofline']['frac']))
except Exception as e:
    print('$VAR=$v: FAILED', l[-300:])
" >> gpurun_out/ab.log
done
cat gpurun_out/ab.log
