#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r06_d2b; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_loopback.py tests/test_dist.py -x -q -m gpu 2>&1 | tail -3
for cfg in "22 8" "26 8" "22 4" "22 2"; do set -- $cfg; timeout 900 python tools/dist2_single.py $1 $2 native 2>/dev/null | grep -v amdgpu.ids | tail -6 > $O/dist2_native_$1_$2.log; tail -3 $O/dist2_native_$1_$2.log | cut -c1-200; done
