#!/bin/bash
# the whole GPU suite, two bench.py runs and a direction-optimising one
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r2b; rm -rf $O; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests -q -x -m gpu --timeout 900 > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
for i in 1 2; do timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-420; done | tee $O/bench.log
timeout 300 python bench.py --mode do --alpha 64 --steps 16 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-300
