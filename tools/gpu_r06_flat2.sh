#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; O=gpurun_out/r06_flat2; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -k "flat_graph or m_launches or per_source or run_many or kernel_variants or degenerate" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 600 python tools/bfs_levels_plain.py --graph uniform --scale 22 --sources 2 --configs ";" 2>&1 | grep -v amdgpu | tail -9
timeout 600 python bench.py --graph uniform --scale 22 --steps 16 --warmup 2 --cpu-seconds 2 > $O/bench_uniform.json 2> $O/err.txt; python -c "
import json; j=json.load(open('$O/bench_uniform.json')); print('uniform-22', j['ms_per_step'], j['value'], j['roofline'].get('whole_bfs_frac'), j.get('parity_vs_oracle'))"
for i in 1 2; do timeout 600 python bench.py --no-cpu-baseline > $O/bench_$i.json 2>> $O/err.txt; python -c "
import json; j=json.load(open('$O/bench_$i.json')); print('rmat-22 batch', j['ms_per_step'], 'per_call', j['per_call']['ms_per_step'])"; done
timeout 600 python bench.py --per-call --steps 64 --no-cpu-baseline > $O/bench_per_call.json 2>> $O/err.txt; python -c "
import json; j=json.load(open('$O/bench_per_call.json')); print('per-call (fresh sources)', j['ms_per_step'])"
timeout 600 python bench.py --graph grid2d --scale 22 --steps 4 --warmup 2 --cpu-seconds 2 > $O/bench_grid.json 2>> $O/err.txt; python -c "
import json; j=json.load(open('$O/bench_grid.json')); print('grid2d-22', j['ms_per_step'], j.get('parity_vs_oracle'))"
