#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
for w in 0 1 2; do
MGX_BFS_WAVE=$w timeout 600 python tools/bfs_levels.py --scale 22 --runs 1 > gpurun_out/levels_wave$w.log 2>&1
echo "wave=$w"; grep -E "src|level  [2345]" gpurun_out/levels_wave$w.log
MGX_BFS_WAVE=$w timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline > gpurun_out/bench_wave$w.log 2>&1
tail -1 gpurun_out/bench_wave$w.log | cut -c1-330
done
