#!/bin/bash
cd $GRAFT_REPO_ROOT
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 600 -x -k "sssp" 2>&1 | tail -2
for l in 0 1; do
echo "== layout $l"
MGX_SSSP_LAYOUT=$l timeout 300 python -u tools/sssp_bench.py --scale 22 --runs 3 2>&1 | grep -v amdgpu.ids | tail -3
done
