#!/bin/bash
# A/B of N builds of the library on one box: bench.py alternating between them, $1 rounds, bench arguments in $BENCH_ARGS
# usage (GPU box): BENCH_ARGS="--mode sssp" bash tools/ab_libs_n.sh 3 mini_amd/libmgx_base.so mini_amd/libmgx.so ...
R=$1; shift
O=gpurun_out/ab_libs_n; rm -rf $O; mkdir -p $O
for i in $(seq 1 $R); do
  for L in "$@"; do
    tag=$(basename $L .so)_$i
    MGX_LIB=$L timeout 300 python bench.py --no-cpu-baseline $BENCH_ARGS > $O/$tag.json 2> $O/$tag.err || echo "$tag failed: $(tail -n 2 $O/$tag.err)"
  done
done
python - "$O" <<'PY'
import json, glob, sys, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        j = json.load(open(f))
    except Exception as e:
        print(os.path.basename(f), "unreadable", e); continue
    print("%-22s ms %.4f  frac %.3f  parity %s" % (os.path.basename(f)[:-5], j["ms_per_step"], j.get("roofline", {}).get("frac", 0), j.get("parity_vs_oracle")))
PY
