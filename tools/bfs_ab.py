#!/usr/bin/env python3
"""A/B of run-time switches of the fused BFS on ONE graph in ONE process: wall time per traversal (same sources for every
configuration, labels compared with the first configuration's).
  python tools/bfs_ab.py --scale 22 --configs "MGX_BFS_DENSE=0,MGX_BFS_CHAIN_MAX_EDGES=0;MGX_BFS_DENSE=0;;" --rounds 2
A configuration is a comma-separated list of VAR=value ("" = defaults); rounds interleave the configurations."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mini_amd
from mini_amd import rmat

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=22)
ap.add_argument("--steps", type=int, default=16)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--rounds", type=int, default=2)
ap.add_argument("--mode", type=int, default=0)
ap.add_argument("--alpha", type=float, default=4.0)
ap.add_argument("--graph", choices=["rmat", "uniform", "grid2d"], default="rmat")
ap.add_argument("--configs", default=";MGX_BFS_DENSE=0;MGX_BFS_CHAIN_MAX_EDGES=0;MGX_BFS_DENSE=0,MGX_BFS_CHAIN_MAX_EDGES=0")
a = ap.parse_args()
ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
g = {"rmat": lambda: rmat.rmat_csr(ctx, a.scale, 16, seed=a.scale), "uniform": lambda: rmat.uniform_csr(ctx, a.scale, 16, seed=a.scale),
     "grid2d": lambda: rmat.grid2d_csr(ctx, a.scale)}[a.graph]()
graph = mini_amd.Graph.from_device(ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"])
graph.build_layout()
ro = g["row_offsets"].cpu().numpy()
srcs = rmat.pick_sources(ro, a.steps + a.warmup, a.scale)
configs = a.configs.split(";")
ref = None
results = {c: [] for c in configs}
touched = set()
for rnd in range(a.rounds):
    for cfg in configs:
        for k in touched:
            os.environ.pop(k, None)
        for kv in [x for x in cfg.split(",") if x]:
            k, v = kv.split("=")
            os.environ[k] = v
            touched.add(k)
        bfs = mini_amd.BfsProblem(graph, srcs[0])       # (the switches are read once per handle: a handle per configuration and round)
        for s in srcs[:a.warmup]:
            bfs.run(s, a.mode, a.alpha)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        timed = [int(s) for s in srcs[a.warmup:]]
        bufs = [bfs.new_stats() for _ in timed]
        t0 = time.perf_counter()
        for s, b in zip(timed, bufs):
            bfs.run_into(s, a.mode, a.alpha, b)          # (bench.py's timed loop)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        sts = [bfs.stats_dict(b) for b in bufs]
        st = sts[-1]
        m_t, nsl, nb = sum(x["m_t"] for x in sts), sum(x["slots"] for x in sts), 0
        lab = bfs.labels()
        if ref is None:
            ref = lab.copy()
        same = bool(np.array_equal(lab, ref))
        results[cfg].append(dt / a.steps * 1e3)
        print("round %d  %-60s %.4f ms/BFS  %.1f GTEPS  slots %.2f dense %d vshort %d lazy %d cold %d small %d  labels_equal %s" % (
            rnd, cfg or "(defaults)", dt / a.steps * 1e3, m_t / dt / 1e9, nsl / a.steps, st["dense_slots"], st["vshort_slots"], st.get("lazy_slots", 0), st.get("cold_slots", 0), st["small_levels"], same), flush=True)
for cfg in configs:
    r = sorted(results[cfg])
    print("best  %-60s %.4f ms/BFS   median %.4f" % (cfg or "(defaults)", r[0], r[len(r) // 2]))
