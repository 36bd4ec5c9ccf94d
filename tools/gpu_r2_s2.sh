#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s2; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_streams.py -q -x -m gpu --timeout 600 -k "sssp or stream" > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
for cfg in "" "MGX_SSSP_SLICED=0" "MGX_SSSP_SLICED=2" "MGX_SSSP_SLICED=6"; do
  for kv in $cfg; do export "$kv"; done
  echo "[$cfg]"; timeout 600 python tools/sssp_bench.py --scale 22 --runs 4 --check 0 2>&1 | grep -E "^SSSP RMAT-22 fused loop:|^fused src"
  for kv in $cfg; do unset "${kv%%=*}"; done
done
