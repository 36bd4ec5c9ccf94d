#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 -x > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline > gpurun_out/bench_fused.log 2>&1
tail -1 gpurun_out/bench_fused.log | cut -c1-200
for G in 1 2 8; do
echo "== dist2 scale 22 G=$G"
timeout 600 python tools/dist2_single.py 22 $G 2>&1 | tail -5
done
echo "== dist2 scale 25 G=8"
timeout 900 python tools/dist2_single.py 25 8 2>&1 | tail -5
