#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -12 gpurun_out/pytest_gpu.log
for sm in 8192 0; do
echo "== small_max $sm"
MGX_BFS_SMALL_MAX_EDGES=$sm timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-210
MGX_BFS_SMALL_MAX_EDGES=$sm timeout 600 python tools/bfs_levels.py --scale 22 --runs 1 2>&1 | grep -E "src|level  [0-9]|slots|batches"
done
