#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu --timeout 600 -x 2>&1 | tail -5
for lm in 0 64; do
echo "== long_min $lm"
MGX_BFS_LONG_MIN=$lm timeout 600 python tools/bfs_levels.py --scale 22 --runs 1 2>&1 | grep -E "src|claims|level  [0-9]" | head -24
MGX_BFS_LONG_MIN=$lm timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-250
done
