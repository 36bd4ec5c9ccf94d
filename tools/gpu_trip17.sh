#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -8 gpurun_out/pytest_gpu.log
for lm in 64 0; do
echo "== long_min $lm"
MGX_BFS_LONG_MIN=$lm timeout 600 python tools/bfs_levels.py --scale 22 --runs 2 2>&1 | grep -E "src|claims|level  [0-9]|stream|wave" | head -40
MGX_BFS_LONG_MIN=$lm timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-900
done
