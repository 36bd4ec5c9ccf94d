#!/bin/bash
# fold tiers of the sliced neighbour-reduce (MGX_NR_FOLD_DEGS=workgroup/wave/eight-lanes) -> gpurun_out/nrs/fold_sweep.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/nrs; mkdir -p $O; rm -f $O/fold_sweep.txt
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "neighbour_reduce or pr_matches" > $O/pytest_nr.log 2>&1; tail -2 $O/pytest_nr.log
for d in ${DEGS:-16384/2048/128 16384/1024/64 8192/512/32 65536/4096/256 16384/2048/2048 16384/128/128}; do
  export MGX_NR_FOLD_DEGS=$d
  SLICED=1 PARTS=3 bash tools/gpu_nrs_parts.sh | sed "s|^|degs=$d |" >> $O/fold_sweep.txt
done
grep fold $O/fold_sweep.txt
