#!/bin/bash
cd $GRAFT_REPO_ROOT
ulimit -c 0
for i in 1 2 3; do
timeout 60 python -u tools/sssp_dbg.py 18 2>&1 | grep -v amdgpu.ids | grep -E "fault|iterations|illegal" | tail -1 | cut -c1-150
done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 600 -x 2>&1 | tail -2
timeout 300 python -u tools/sssp_bench.py --scale 22 --runs 3 2>&1 | grep -v amdgpu.ids | tail -6
timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-210
