#!/bin/bash
ulimit -c 0
O=gpurun_out/r2i; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu --timeout 900 > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -8 $O/pytest_gpu.log
timeout 600 python tools/sssp_bench.py --scale 22 --runs 4 --check 0 > $O/sssp.log 2>&1
echo "sssp rc=$?"; grep -E "^SSSP" $O/sssp.log
for a in 0.05 1 4 16 64; do timeout 300 python bench.py --mode do --alpha $a --steps 16 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('DO alpha=$a: %.4f ms  %.1f GTEPS  parity %s' % (j['ms_per_step'], j['value']/1e3, j.get('parity_vs_oracle')))"; done
