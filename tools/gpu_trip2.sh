#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -40 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
for ept in 4 8; do
MGX_BFS_EPT=$ept timeout 600 python tools/bfs_levels.py --scale 22 --runs 2 > gpurun_out/levels_ept$ept.log 2>&1
echo "levels ept=$ept rc=$?"; tail -12 gpurun_out/levels_ept$ept.log
MGX_BFS_EPT=$ept timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline > gpurun_out/bench_s22_ept$ept.log 2>&1
echo "bench22 ept=$ept rc=$?"; tail -1 gpurun_out/bench_s22_ept$ept.log | cut -c1-400
done
