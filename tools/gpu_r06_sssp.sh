#!/bin/bash
# round 6: the fused SSSP's sweep gated by frontier activity -- parity, then A/B against the library of the commit before
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_sssp; rm -rf $O; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -k "sssp" > $O/pytest_sssp.txt 2>&1; tail -3 $O/pytest_sssp.txt
for i in 1 2; do
  for L in base ""; do
    MGX_LIB=$R/mini_amd/libmgx${L:+_$L}.so timeout 300 python bench.py --mode sssp --no-cpu-baseline > $O/bench_sssp_${L:-new}_$i.json 2> $O/err.txt || tail -3 $O/err.txt
    python - $O/bench_sssp_${L:-new}_$i.json <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j.get("roofline",{})
print(sys.argv[1].split("/")[-1], "ms %.4f value %.0f frac %s parity %s" % (j["ms_per_step"], j["value"], r.get("frac"), j.get("parity_vs_oracle")))
PY
  done
done
for L in base ""; do
  echo "=== iterations [${L:-new}]"; MGX_LIB=$R/mini_amd/libmgx${L:+_$L}.so timeout 300 python tools/sssp_iterations.py 2>&1 | grep -v amdgpu.ids
done > $O/iterations.txt; cat $O/iterations.txt
