R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/j21; rm -rf $O; mkdir -p $O; cd $R
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -m gpu -x --timeout 900 -k "bfs or many or batch or config" > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -1
for i in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline > $O/bench$i.log 2>&1; tail -1 $O/bench$i.log | python3 -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']; print(j['value'], j['ms_per_step'], 'per_call', j['per_call']['ms_per_step'], 'build_us', r['build_us_per_traversal'], 'push_us', r['push_us_per_traversal'], 'parity', j.get('parity_vs_oracle'), 'reruns', j.get('batch_reruns'))"; done
bash tools/gpu_timeline.sh "" 2>&1 | sed -n 18,34p
