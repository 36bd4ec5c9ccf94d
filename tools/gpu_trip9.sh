#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
export MGX_BFS_HOT_SHAPE=1
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu.log
for alpha in 2 4 16; do
timeout 600 python tools/bfs_levels.py --scale 22 --runs 1 --mode 1 --alpha $alpha > gpurun_out/levels_do_a$alpha.log 2>&1
echo "DO alpha=$alpha"; grep -E "src|claims|level " gpurun_out/levels_do_a$alpha.log | head -12
done
timeout 600 python bench.py --steps 16 --warmup 2 --mode do --alpha 4 > gpurun_out/bench_do.log 2>&1
echo "bench do rc=$?"; tail -1 gpurun_out/bench_do.log
timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline > gpurun_out/bench_push.log 2>&1
echo "bench push rc=$?"; tail -1 gpurun_out/bench_push.log | cut -c1-300
