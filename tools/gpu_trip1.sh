#!/bin/bash
# first GPU trip: microbench, parity tests, smoke, small + full bench
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
rocminfo | grep -E "Marketing|Compute Unit|gfx" | head -8 > gpurun_out/rocminfo.txt 2>&1
nproc > gpurun_out/host.txt; lscpu | grep "Model name" >> gpurun_out/host.txt
timeout 300 ./tools/microbench > gpurun_out/microbench.jsonl 2>&1
echo "microbench rc=$?"
timeout 1500 python -m pytest tests -q -m gpu -x --timeout 600 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"
tail -30 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
timeout 600 python bench.py --scale 18 --steps 8 --warmup 2 > gpurun_out/bench_s18.log 2>&1
echo "bench18 rc=$?"; tail -2 gpurun_out/bench_s18.log
timeout 900 python bench.py --steps 16 --warmup 2 > gpurun_out/bench_s22.log 2>&1
echo "bench22 rc=$?"; tail -2 gpurun_out/bench_s22.log
