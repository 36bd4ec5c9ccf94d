#!/bin/bash
# A/B of mini_amd/libmgx_base.so against mini_amd/libmgx.so on the default bench (ab_libs.sh) + a parity subset on the new library
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_ab; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bfs and not large" > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
bash tools/ab_libs.sh mini_amd/libmgx_base.so mini_amd/libmgx.so ${1:-3} > $O/ab.txt 2>&1; cat $O/ab.txt
