#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 600 -x 2>&1 | tail -3
for sm in 0 256 1024 2048 8192; do
echo "== small_max $sm"
MGX_BFS_SMALL_MAX_EDGES=$sm timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-210
done
MGX_BFS_SMALL_MAX_EDGES=1024 timeout 600 python tools/bfs_levels.py --scale 22 --runs 1 2>&1 | grep -E "src|level  [0-9]|slots|batches"
