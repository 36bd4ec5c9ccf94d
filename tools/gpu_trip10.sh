#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "sssp or fixtures" > gpurun_out/pytest_sssp.log 2>&1; tail -2 gpurun_out/pytest_sssp.log
timeout 1200 python tools/sssp_bench.py --scale 22 --runs 3 > gpurun_out/sssp_s22.log 2>&1; tail -5 gpurun_out/sssp_s22.log
