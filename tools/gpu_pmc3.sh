#!/bin/bash
# PMC passes over the product push kernel and the queue build of single traversals (round 4): per DISPATCH rows of the
# largest launches -- instruction mix, LDS and memory wait shares -- to tell issue-bound from bandwidth-bound.
# usage (GPU box): bash tools/gpu_pmc3.sh        -> gpurun_out/pmc3/summary.txt
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc3; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ulimit -c 0
rm -rf $O/p*
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVES SQ_LEVEL_WAVES" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TA_TCP_STATE_READ_sum TCP_GATE_EN1_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --per-call > $O/p$i.log 2>&1
  echo "pmc set $i rc=$?"
done
cd $R
python3 - <<'PY' > $O/summary.txt
import csv,glob,collections,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc3'
rows=collections.defaultdict(dict)   # (kernel short, dispatch id) -> counter -> value
for f in sorted(glob.glob(O+'/p*/**/*counter_collection.csv', recursive=True)):
    pset=f.split('/pmc3/')[1].split('/')[0]
    for r in csv.DictReader(open(f)):
        kn=r['Kernel_Name']
        short=None
        for s in ('k_bfs_push<false, 0>','k_bfs_build2','k_bfs_mini','k_bfs_chain_inplace','k_bfs_fused_init'):
            if s in kn: short=s
        if not short: continue
        rows[(short,pset,int(r['Dispatch_Id']))][r['Counter_Name']]=float(r['Counter_Value'])
# per kernel and counter set: the dispatches ordered by their first counter, the top 4 printed
by=collections.defaultdict(list)
for (short,pset,d),c in rows.items(): by[(short,pset)].append((d,c))
for (short,pset) in sorted(by):
    lst=by[(short,pset)]
    key=sorted(lst[0][1])[0]
    for k in ('SQ_WAVE_CYCLES','SQ_ACTIVE_INST_VALU','SQ_ACTIVE_INST_VMEM','GRBM_GUI_ACTIVE','TCP_PENDING_STALL_CYCLES_sum'):
        if k in lst[0][1]: key=k; break
    lst.sort(key=lambda x:-x[1].get(key,0))
    tot=collections.Counter()
    for d,c in lst:
        for k,v in c.items(): tot[k]+=v
    print('== %s  [%s]  %d dispatches; totals: %s'%(short,pset,len(lst),' '.join('%s=%.4g'%(k,tot[k]) for k in sorted(tot))))
    for d,c in lst[:3]:
        print('   dispatch %6d  %s'%(d,' '.join('%s=%.4g'%(k,c[k]) for k in sorted(c))))
PY
cat $O/summary.txt | cut -c1-400
