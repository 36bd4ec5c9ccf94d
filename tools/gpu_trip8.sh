#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 600 -x -k "hub_first or large or rmat_parity_all" > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu.log
for shape in 0 1 2; do
MGX_BFS_HOT_SHAPE=$shape timeout 600 python tools/bfs_levels.py --scale 22 --runs 1 > gpurun_out/levels_shape$shape.log 2>&1
echo "shape=$shape"; grep -E "src|level  [234]" gpurun_out/levels_shape$shape.log
MGX_BFS_HOT_SHAPE=$shape timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline > gpurun_out/bench_shape$shape.log 2>&1
tail -1 gpurun_out/bench_shape$shape.log | cut -c1-160
done
