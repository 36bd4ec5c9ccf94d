#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r06_flat; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_configs.py -x -q -k "flat_graph" > $O/pytest_flat.txt 2>&1; tail -12 $O/pytest_flat.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "bfs and not large" > $O/pytest_bfs.txt 2>&1; tail -2 $O/pytest_bfs.txt
for sw in 1 0; do
  MGX_BFS_FLAT_LISTS=$sw timeout 600 python bench.py --graph uniform --scale 22 --steps 16 --warmup 2 --cpu-seconds 2 > $O/bench_uniform_$sw.json 2> $O/err.txt || tail -3 $O/err.txt
  python - $O/bench_uniform_$sw.json $sw <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j.get("roofline",{})
print("uniform-22 flat_lists=%s ms %.4f value %.0f whole %s bytes %d parity %s" % (sys.argv[2], j["ms_per_step"], j["value"], r.get("whole_bfs_frac"), j["layout"]["device_bytes"], j.get("parity_vs_oracle")))
PY
done
timeout 600 python bench.py --per-call --steps 64 --no-cpu-baseline > $O/bench_per_call.json 2>> $O/err.txt; python -c "
import json; j=json.load(open('$O/bench_per_call.json')); print('per-call', j['ms_per_step'])"
timeout 600 python bench.py --no-cpu-baseline > $O/bench.json 2>> $O/err.txt; python -c "
import json; j=json.load(open('$O/bench.json')); print('batch', j['ms_per_step'], j['per_call']['ms_per_step'])"
