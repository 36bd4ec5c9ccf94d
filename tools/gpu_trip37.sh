#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_dist.py -q -m gpu --timeout 600 -x 2>&1 | tail -1
for nt in 1024 512 256; do
echo "== build nt $nt"
MGX_BFS_BUILD_NT=$nt timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-210
MGX_BFS_BUILD_NT=$nt timeout 600 python tools/bfs_levels.py --scale 22 --runs 1 2>&1 | grep -E "level  [1-5]"
done
