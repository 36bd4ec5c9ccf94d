#!/bin/bash
# PMC passes over the push bench: instruction mix and stall reasons of the push kernels and the queue build
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
export MGX_BFS_MERGED_PUSH=${MGX_BFS_MERGED_PUSH:-0}
ulimit -c 0
rm -rf $O/p*
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_LEVEL_WAVES"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --output-format csv -d $O/p$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check > $O/p$i.log 2>&1
  echo "pmc set $i rc=$?"
done
cd $R
python3 - <<'PY'
import csv,glob,collections,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc2'
for f in sorted(glob.glob(O+'/p*/**/*counter_collection.csv', recursive=True)):
    agg=collections.Counter(); cnt=collections.Counter(); mx=collections.Counter()
    for r in csv.DictReader(open(f)):
        for kn in ('push_level_stream','push_level_wave','k_bfs_build'):
            if kn in r['Kernel_Name']:
                agg[(kn,r['Counter_Name'])]+=float(r['Counter_Value']); cnt[(kn,r['Counter_Name'])]+=1; mx[(kn,r['Counter_Name'])]=max(mx[(kn,r['Counter_Name'])],float(r['Counter_Value']))
    for k in sorted(agg): print('%-18s %-24s total=%.4g dispatches=%d max=%.4g'%(k[0],k[1],agg[k],cnt[k],mx[k]))
PY
