#!/usr/bin/env python3
"""HBM traffic per KERNEL from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh (gpurun_out/round/pmc_*): for every
mode (BFS push, --mode sssp, --mode pr) the bytes each kernel moved per bench step -- 2 x FETCH_SIZE + WRITE_SIZE, the gfx950
correction of MI355X_MICROARCH.md -- so that the gap between a mode's counter traffic and its algorithmic bytes has an owner.
   python tools/pmc_by_kernel.py <round dir> [steps incl. warm-up of the pmc commands = 5]"""
import collections, csv, glob, os, sys
O = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5          # --steps 4 --warmup 1
def short(n):
    n = n.replace("void mgx::", "").replace("mgx::", "").replace("gunrock::", "")
    return n.split("(")[0][:64]
for mode, tag in (("push", ""), ("sssp", "sssp_"), ("pr", "pr_")):
    tot = collections.defaultdict(lambda: {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "n": 0})
    for cn in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(O, "pmc_%s%s" % (tag, cn), "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] != cn:
                    continue
                k = short(r["Kernel_Name"])
                if not (k.startswith("k_") or "k_" in k):
                    continue
                tot[k][cn] += float(r["Counter_Value"]) * 1024.0
                if cn == "FETCH_SIZE":
                    tot[k]["n"] += 1
    if not tot:
        continue
    rows = sorted(((2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"], k, v) for k, v in tot.items()), reverse=True)
    allb = sum(b for b, _, _ in rows)
    print("== mode %s: %.1f MB per step over all kernels (2 x FETCH + WRITE; %d steps incl. warm-up and the bench's extra passes count as steps of their own)" % (mode, allb / steps / 1e6, steps))
    for b, k, v in rows[:12]:
        print("   %-64s %6d dispatches  read x2 %9.1f MB  written %8.1f MB  = %9.1f MB per dispatch  %5.1f %% of the mode's traffic" % (
            k, v["n"], 2.0 * v["FETCH_SIZE"] / max(v["n"], 1) / 1e6, v["WRITE_SIZE"] / max(v["n"], 1) / 1e6, b / max(v["n"], 1) / 1e6, 100.0 * b / max(allb, 1.0)))
