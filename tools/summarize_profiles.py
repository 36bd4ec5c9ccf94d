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
        for kn in ("k_bfs_push<false, 0>", "k_bfs_push<false, 1>", "k_bfs_push<false, 2>", "k_bfs_push<false, 3>", "k_bfs_build", "k_bfs_fused_init", "k_bfs_pull_level", "k_bfs_chain_inplace", "k_bfs_mini", "k_bfs_publish"):
            if kn in r["Name"]:
                out[kn] = {"calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]), "total_ns": int(r["TotalDurationNs"])}
                print("kernel-trace: %s calls=%s avg=%.1f us total=%.3f ms" % (kn, r["Calls"], float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 1e6))
pm = {}
KERNEL = "k_bfs_push<false, 0>"          # the product kernel: one launch per slot (bench.py's roofline.kernel)
for f in glob.glob(os.path.join(O, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    agg, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(f)):
        if KERNEL in r["Kernel_Name"]:
            agg[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
    for k in agg:
        pm[k] = {"sum": agg[k], "dispatches": cnt[k], "per_dispatch": agg[k] / cnt[k]}
        print("pmc %-22s dispatches=%d per_dispatch=%.6g" % (k, cnt[k], agg[k] / cnt[k]))
out["pmc"] = pm
if "FETCH_SIZE" in pm and "WRITE_SIZE" in pm:
    # rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB.  MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE is
    # TCC_EA0_RDREQ x 64 B while the requests are 128 B -- HALF of the streamed bytes, doubled here.  Round 1's own data
    # shows the same for this kernel's 4-byte-per-lane stream (raw FETCH below the compulsory col_indices bytes).
    # WRITE_SIZE is exact.
    fetch_raw = pm["FETCH_SIZE"]["per_dispatch"] * 1024.0
    write_raw = pm["WRITE_SIZE"]["per_dispatch"] * 1024.0
    out["hbm_bytes_per_launch_raw"] = fetch_raw + write_raw
    out["hbm_bytes_per_launch"] = 2.0 * fetch_raw + write_raw
    print("HBM bytes per launch: FETCH raw %.4g (x2 = %.4g)  WRITE %.4g  -> corrected %.4g" % (fetch_raw, 2 * fetch_raw, write_raw, 2 * fetch_raw + write_raw))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    scale, mode = 22, "push"
    for f in glob.glob(os.path.join(O, "bench.log")):
        line = [l for l in open(f) if l.startswith("{")]
        if line:
            j = json.loads(line[-1]); scale = j["config"]["scale"]
    traffic = {"kernel": KERNEL, "scale": scale, "mode": mode, "source_sha": bench.source_sha(),
               "fetch_size_kib_per_dispatch_raw": pm["FETCH_SIZE"]["per_dispatch"], "write_size_kib_per_dispatch_raw": pm["WRITE_SIZE"]["per_dispatch"],
               "dispatches": pm["FETCH_SIZE"]["dispatches"],
               "fetch_bytes_corrected": 2.0 * fetch_raw, "write_bytes": write_raw, "hbm_bytes_per_launch": 2.0 * fetch_raw + write_raw,
               "correction": "2 x FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM: gfx950 FETCH_SIZE counts 128-B read requests as 64 B)",
               "command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-check; "
                          "average over every dispatch of the kernel (all slots of all traversals, as bench.py's alg_bytes_per_launch)"}
    entries = [traffic]
    # the other modes' dominant kernels: the same two counters from passes over `bench.py --mode sssp` / `--mode pr`
    # (kernels summed, the kernel whose dispatches count the "launches" bench.py divides by)
    for mode, pat, unit_pat, kname in (("sssp", "k_sssp_relax", "k_sssp_relax<", "k_sssp_relax<1024> (+ k_sssp_relax_dense<1024> on heavy iterations)"),
                                       ("pr", "k_nr", "k_nr_values", "neighbour-reduce operator")):
        vals = {}
        for cn in ("FETCH_SIZE", "WRITE_SIZE"):
            agg, cnt = 0.0, 0
            for f in glob.glob(os.path.join(O, "pmc_%s_%s" % (mode, cn), "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if pat in r["Kernel_Name"] and r["Counter_Name"] == cn and "k_nrs_counts" not in r["Kernel_Name"] and "k_nrs_fill" not in r["Kernel_Name"]:   # (the one-time builders of the sliced long rows are not the operator)
                        agg += float(r["Counter_Value"])
                        if unit_pat in r["Kernel_Name"]:
                            cnt += 1
            if cnt:
                vals[cn] = (agg, cnt)
        if len(vals) == 2:
            # per "launch" as bench.py counts them: sssp = one relax launch per iteration (the sweep kernel behind it rides on it),
            # pr = one operator call (every kernel of it)
            disp = vals["FETCH_SIZE"][1]
            fr = vals["FETCH_SIZE"][0] * 1024.0 / disp
            wr = vals["WRITE_SIZE"][0] * 1024.0 / vals["WRITE_SIZE"][1]
            entries.append({"kernel": kname, "scale": scale, "mode": mode, "source_sha": bench.source_sha(), "dispatches": disp,
                            "fetch_bytes_corrected": 2.0 * fr, "write_bytes": wr, "hbm_bytes_per_launch": 2.0 * fr + wr,
                            "correction": traffic["correction"],
                            "command": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --mode %s --steps 4 --warmup 1 --no-cpu-baseline --no-check; dispatches matching '%s'" % (mode, pat)})
            print("pmc %s: HBM bytes per launch %.4g (read x2 %.4g, written %.4g)" % (mode, 2 * fr + wr, 2 * fr, wr))
    out_json = dict(traffic)
    out_json["entries"] = entries
    json.dump(out_json, open(os.path.join(O, "pmc_traffic.json"), "w"), indent=1)
json.dump(out, open(os.path.join(O, "summary.json"), "w"), indent=1)
for f in glob.glob(os.path.join(O, "bench.log")):
    line = [l for l in open(f) if l.startswith("{")]
    if line:
        j = json.loads(line[-1]); print("bench: value=%s MTEPS ms/step=%s roofline=%s cpu=%s parity=%s" % (j["value"], j["ms_per_step"], j["roofline"], j["cpu_baseline"], j.get("parity_vs_oracle")))
