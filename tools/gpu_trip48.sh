#!/bin/bash
cd $GRAFT_REPO_ROOT
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_dropin.py -q -m gpu --timeout 600 -x -k "sssp or fixture or dropin or kcore or reference" 2>&1 | tail -2
for lm in 64 0 16; do
echo "== sssp long_min $lm"
MGX_SSSP_LONG_MIN=$lm timeout 300 python -u tools/sssp_bench.py --scale 22 --runs 3 2>&1 | grep -v amdgpu.ids | tail -3
done
