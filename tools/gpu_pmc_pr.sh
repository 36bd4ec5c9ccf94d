#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes of `bench.py --mode pr` alone -> the "pr" entry of profiles/pmc_traffic.json (the other entries stay),
# then the bench line that reports it; and the bench lines of the graphs GRAPHS names -> gpurun_out/pmc_pr/
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_pr; rm -rf $O; mkdir -p $O/keep
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/pmc_pr_$c -- python3 $R/bench.py --mode pr --steps 4 --warmup 1 --no-cpu-baseline --no-check > $O/pmc_pr_$c.log 2>&1
done
cd $R
python3 - <<'PY'
import csv, glob, json, os, sys
R = os.environ["GRAFT_REPO_ROOT"]; O = R + "/gpurun_out/pmc_pr"
sys.path.insert(0, R)
import bench
vals = {}
per = {}
for cn in ("FETCH_SIZE", "WRITE_SIZE"):
    agg, cnt = 0.0, 0
    for f in glob.glob(os.path.join(O, "pmc_pr_" + cn, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_nr" in k and r["Counter_Name"] == cn and "k_nrs_counts" not in k and "k_nrs_fill" not in k:
                agg += float(r["Counter_Value"])
                per.setdefault(k.split("<")[0].split("::")[-1], {}).setdefault(cn, [0.0, 0])
                per[k.split("<")[0].split("::")[-1]][cn][0] += float(r["Counter_Value"]); per[k.split("<")[0].split("::")[-1]][cn][1] += 1
                if "k_nr_values" in k:
                    cnt += 1
    vals[cn] = (agg, cnt)
disp = vals["FETCH_SIZE"][1]
fr = vals["FETCH_SIZE"][0] * 1024.0 / disp
wr = vals["WRITE_SIZE"][0] * 1024.0 / vals["WRITE_SIZE"][1]
entry = {"kernel": "neighbour-reduce operator", "scale": 22, "mode": "pr", "source_sha": bench.source_sha(), "dispatches": disp,
         "fetch_bytes_corrected": 2.0 * fr, "write_bytes": wr, "hbm_bytes_per_launch": 2.0 * fr + wr,
         "correction": "2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM: gfx950 FETCH_SIZE counts 128-B read requests as 64 B)",
         "command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --mode pr --steps 4 --warmup 1 --no-cpu-baseline --no-check; "
                    "every dispatch of k_nr_values / k_nrs_edges / k_nrs_fold (k_nr_edges / k_nr_fold), per operator call; the one-time builders k_nrs_counts / k_nrs_fill left out (tools/gpu_pmc_pr.sh)",
         "per_kernel_MB": {k: round((2.0 * v["FETCH_SIZE"][0] / v["FETCH_SIZE"][1] + v["WRITE_SIZE"][0] / v["WRITE_SIZE"][1]) * 1024.0 / 1e6, 1) for k, v in per.items() if "FETCH_SIZE" in v and "WRITE_SIZE" in v}}
p = R + "/profiles/pmc_traffic.json"
j = json.load(open(p))
j["entries"] = [e for e in j.get("entries", []) if e.get("mode") != "pr"] + [entry]
json.dump(j, open(p, "w"), indent=1)
json.dump(j, open(O + "/keep/pmc_traffic.json", "w"), indent=1)
print("pr: HBM bytes per call %.4g (read x2 %.4g, written %.4g); per kernel %s" % (2 * fr + wr, 2 * fr, wr, entry["per_kernel_MB"]))
PY
rm -rf $O/pmc_pr_FETCH_SIZE $O/pmc_pr_WRITE_SIZE
timeout 600 python bench.py --mode pr > $O/bench_pr.log 2>&1
grep '^{' $O/bench_pr.log | tail -1 > $O/keep/bench_line_pr.json; cut -c1-400 $O/keep/bench_line_pr.json
IFS=';' read -ra GRAPH_LIST <<< "${GRAPHS:-}"
for g in "${GRAPH_LIST[@]}"; do
  set -- $g
  timeout 600 python bench.py --graph $1 --scale $2 --steps $3 --warmup 2 --cpu-seconds 5 > $O/bench_$1_$2.log 2>&1
  grep '^{' $O/bench_$1_$2.log | tail -1 > $O/keep/bench_line_$1_$2.json; cut -c1-200 $O/keep/bench_line_$1_$2.json
done
