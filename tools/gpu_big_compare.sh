#!/bin/bash
# kernel time of the fused engine on RMAT-25 beside the two-shard run of RMAT-26 (a shard is as large as RMAT-25)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in "25 0" "26 0"; do set -- $c
  rm -rf /tmp/tr_$1
  timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$1 -- python3 $R/bench.py --scale $1 --steps 8 --warmup 2 --no-cpu-baseline --no-check > $R/gpurun_out/big_$1.json 2> $R/gpurun_out/big_$1.err
  echo "== scale $1"; python3 - <<PY
import csv,glob,json
f=glob.glob('/tmp/tr_$1/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:9]: print('  %-70s calls %6s total %10.1f us avg %8.1f us' % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs'])/1e3, float(r['AverageNs'])/1e3))
d=json.loads(open('$R/gpurun_out/big_$1.json').read().strip().splitlines()[-1]); print('  ms_per_step', d['ms_per_step'], 'value', d['value'])
PY
done
