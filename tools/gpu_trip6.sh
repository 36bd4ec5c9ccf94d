#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu.log
for cfg in "4 5" "4 6" "4 4" "8 4" "8 3"; do
set -- $cfg
MGX_BFS_EPT=$1 MGX_BFS_OCC=$2 timeout 600 python tools/bfs_levels.py --scale 22 --runs 1 > gpurun_out/levels_ept$1_occ$2.log 2>&1
echo "levels ept=$1 occ=$2 rc=$?"; grep -E "src|level  [234]" gpurun_out/levels_ept$1_occ$2.log
MGX_BFS_EPT=$1 MGX_BFS_OCC=$2 timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline > gpurun_out/bench_s22_ept$1_occ$2.log 2>&1
echo "bench22 ept=$1 occ=$2 rc=$?"; tail -1 gpurun_out/bench_s22_ept$1_occ$2.log | cut -c1-160
done
MGX_BFS_DIAG=1 timeout 600 python tools/bfs_levels.py --scale 22 --runs 1 > gpurun_out/levels_diag.log 2>&1
tail -4 gpurun_out/levels_diag.log
