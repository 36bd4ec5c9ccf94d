#!/usr/bin/env python3
"""Condense a tools/profile_round.sh output directory into the small files that go under profiles/."""
import collections, csv, glob, json, os, sys
O = sys.argv[1]
out = {}
for f in glob.glob(os.path.join(O, "trace", "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    keep = [r for r in rows if "mgx" in r["Name"] or float(r["Percentage"]) > 1.0]
    with open(os.path.join(O, "kernel_stats_mgx.csv"), "w") as g:
        w = csv.DictWriter(g, fieldnames=rows[0].keys()); w.writeheader()
        for r in keep:
            r = dict(r); r["Name"] = r["Name"][:120]; w.writerow(r)
    for r in rows:
        for kn in ("k_bfs_push<false, 0>", "k_bfs_push<false, 1>", "k_bfs_push<false, 2>", "k_bfs_push<false, 3>", "k_bfs_build", "k_bfs_fused_init", "k_bfs_pull_level"):
            if kn in r["Name"]:
                out[kn] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "total_ns": int(r["TotalDurationNs"])}
                print("kernel-trace: %s calls=%s avg=%.1f us total=%.3f ms" % (kn, r["Calls"], float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e6))
pm = {}
for f in glob.glob(os.path.join(O, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    agg, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(f)):
        if "k_bfs_push_level_stream" in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    for k in agg:
        pm[k] = {"sum": agg[k], "dispatches": cnt[k], "per_dispatch": agg[k] / cnt[k]}
        print("pmc %-22s dispatches=%d per_dispatch=%.6g" % (k, cnt[k], agg[k] / cnt[k]))
out["pmc"] = pm
if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
    # rocprofv3 reports KiB-ish units (x1024 B per MI355X_MICROARCH.md / cdna guide section 7); FETCH_SIZE is NOT
    # doubled here: the 2x correction of the guide is calibrated for 16 B/lane streams only, this kernel reads 4 B/lane
    raw = (pm["FETCH_SIZE"]["per_dispatch"] + pm["WRITE_SIZE"]["per_dispatch"]) * 1024.0
    out["hbm_bytes_per_launch_raw"] = raw
    print("HBM bytes per launch (FETCH+WRITE, x1024, uncorrected): %.4g" % raw)
json.dump(out, open(os.path.join(O, "summary.json"), "w"), indent=1)
for f in glob.glob(os.path.join(O, "bench.log")):
    line = [l for l in open(f) if l.startswith("{")]
    if line:
        j = json.loads(line[-1]); print("bench: value=%s MTEPS ms/step=%s roofline=%s cpu=%s parity=%s" % (j["value"], j["ms_per_step"], j["roofline"], j["cpu_baseline"], j.get("parity_vs_oracle")))
