#!/bin/bash
cd $GRAFT_REPO_ROOT
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_dropin.py -q -m gpu --timeout 600 -x -k "reduce or pr or dropin or segreduce or neighbour" 2>&1 | tail -3
for g in 2 1; do
echo "== segreduce gen $g"
MGX_SEGREDUCE_GEN=$g timeout 600 python -u tools/pr_bench.py 2>&1 | grep -v amdgpu.ids | grep -E "segreduce|err|pr enact" | tail -8
done
