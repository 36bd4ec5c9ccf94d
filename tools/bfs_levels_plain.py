#!/usr/bin/env python3
"""Per-level times of the PRODUCT launches (merged push, no events on the stream) from the device-side stamps a level's opener
takes, for several configurations of run-time switches side by side: the median over --reps traversals of each source.
  python tools/bfs_levels_plain.py --scale 22 --sources 3 --configs ";MGX_BFS_DEFER=0" [--graph uniform]
A level's time runs from its opener to the next level's opener: push + build (+ whatever sits between them)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mini_amd
from mini_amd import rmat

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=22)
ap.add_argument("--sources", type=int, default=3)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--configs", default=";MGX_BFS_DEFER=0")
ap.add_argument("--graph", choices=["rmat", "uniform", "grid2d"], default="rmat")
a = ap.parse_args()
ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
g = {"rmat": lambda: rmat.rmat_csr(ctx, a.scale, 16, seed=a.scale), "uniform": lambda: rmat.uniform_csr(ctx, a.scale, 16, seed=a.scale),
     "grid2d": lambda: rmat.grid2d_csr(ctx, a.scale)}[a.graph]()
graph = mini_amd.Graph.from_device(ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"])
graph.build_layout()
ro = g["row_offsets"].cpu().numpy()
srcs = [int(s) for s in rmat.pick_sources(ro, a.sources + 1, a.scale)]
touched = set()
ref = {}
for cfg in a.configs.split(";"):
    for k in touched:
        os.environ.pop(k, None)
    for kv in [x for x in cfg.split(",") if x]:
        k, v = kv.split("=")
        os.environ[k] = v
        touched.add(k)
    bfs = mini_amd.BfsProblem(graph, srcs[0])
    for s in srcs:                       # warm-up: the launch plan learns the graph
        bfs.run(s)
    print("=== [%s]" % (cfg or "defaults"))
    for s in srcs[1:]:
        rows = []
        for _ in range(a.reps):
            st = bfs.run(s)
            rows.append(bfs.level_times_ms())
        lab = bfs.labels()
        if s not in ref:
            ref[s] = lab.copy()
        tr = bfs.level_trace()
        L = min(len(r) for r in rows)
        med = [float(np.median([r[i] for r in rows])) for i in range(L)]
        print("src %d: levels %d reached %d  sum of levels %.1f us  labels_equal %s" % (s, st["levels"], st["reached"], sum(med) * 1e3, bool(np.array_equal(lab, ref[s]))))
        for lv in range(L):
            nf, ne = tr[lv]
            print("  level %2d  nf %9d  edges %10d  %7.1f us" % (lv, nf, ne, med[lv] * 1e3))
    bfs.close()
