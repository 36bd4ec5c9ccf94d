#!/bin/bash
mkdir -p gpurun_out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "reduce or pr_matches" > gpurun_out/pytest_red.log 2>&1; tail -3 gpurun_out/pytest_red.log
timeout 900 python tools/pr_bench.py --scale 22 > gpurun_out/pr_s22.log 2>&1; tail -5 gpurun_out/pr_s22.log
