#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu --timeout 600 > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
timeout 300 python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?"; tail -2 gpurun_out/smoke.log
MGX_BFS_EPT=4 timeout 600 python tools/bfs_levels.py --scale 22 --runs 1 > gpurun_out/levels_ept4.log 2>&1
tail -12 gpurun_out/levels_ept4.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/rocprof_counters.txt 2>&1
echo "counters listed: $(wc -l < $R/gpurun_out/rocprof_counters.txt)"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-check > $R/gpurun_out/prof_trace.log 2>&1
echo "trace rc=$?"; tail -1 $R/gpurun_out/prof_trace.log | cut -c1-300
for set in "TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE"; do
  tag=$(echo $set | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/prof_pmc_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check > $R/gpurun_out/prof_pmc_$tag.log 2>&1
  echo "pmc $tag rc=$?"
done
cd $R
find gpurun_out/prof_trace gpurun_out/prof_pmc_* -name "*.csv" | head -40
du -sh gpurun_out
