#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
timeout 600 python tools/bfs_levels.py --scale 22 --runs 1 2>&1 | grep -E "level  [0-9]|stream|wave" | head -12
bash tools/gpu_trace.sh 2>&1 | grep -E "mgx|MTEPS|fillBuffer"
