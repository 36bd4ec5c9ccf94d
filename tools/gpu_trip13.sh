#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -25 gpurun_out/pytest_gpu.log
for sh in 0 1; do
MGX_CB_SHAPE=$sh timeout 600 python tools/bfs_levels.py --scale 22 --runs 1 > gpurun_out/levels_cb$sh.log 2>&1
echo "cb shape=$sh"; grep -E "src|claims|level  [0-9]" gpurun_out/levels_cb$sh.log
MGX_CB_SHAPE=$sh timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline > gpurun_out/bench_cb$sh.log 2>&1
tail -1 gpurun_out/bench_cb$sh.log | cut -c1-330
done
MGX_BFS_ENGINE=fused timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline > gpurun_out/bench_fused.log 2>&1
tail -1 gpurun_out/bench_fused.log | cut -c1-200
timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline --mode do --alpha 1000 > gpurun_out/bench_cb_do.log 2>&1
tail -1 gpurun_out/bench_cb_do.log | cut -c1-200
