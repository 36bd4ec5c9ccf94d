cd $GRAFT_REPO_ROOT
for s in 24 25 22; do timeout 900 python bench.py --mode pr --scale $s --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | grep '^{' | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('PR RMAT-$s: %.0f MTEPS, %.4f ms, parity %s' % (j['value'], j['ms_per_step'], j.get('parity_vs_oracle')))"; done
