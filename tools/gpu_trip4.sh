#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
for ept in 4 8; do
MGX_BFS_EPT=$ept timeout 600 python tools/bfs_levels.py --scale 22 --runs 1 > gpurun_out/levels_ept$ept.log 2>&1
echo "levels ept=$ept rc=$?"; tail -11 gpurun_out/levels_ept$ept.log
MGX_BFS_EPT=$ept timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline > gpurun_out/bench_s22_ept$ept.log 2>&1
echo "bench22 ept=$ept rc=$?"; tail -1 gpurun_out/bench_s22_ept$ept.log | cut -c1-200
done
