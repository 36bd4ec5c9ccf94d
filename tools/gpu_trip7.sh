#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu.log
timeout 600 python tools/bfs_levels.py --scale 22 --runs 1 > gpurun_out/levels_hot.log 2>&1
echo "levels rc=$?"; tail -11 gpurun_out/levels_hot.log
timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline > gpurun_out/bench_hot.log 2>&1
echo "bench rc=$?"; tail -1 gpurun_out/bench_hot.log | cut -c1-250
timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline --no-layout > gpurun_out/bench_nolayout.log 2>&1
echo "bench nolayout rc=$?"; tail -1 gpurun_out/bench_nolayout.log | cut -c1-250
