#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
for sh in 0 1 2; do
echo "== wave shape $sh"
MGX_BFS_WAVE_SHAPE=$sh timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-210
MGX_BFS_WAVE_SHAPE=$sh timeout 600 python tools/bfs_levels.py --scale 22 --runs 1 2>&1 | grep -E "level  [3-4]|wave"
done
