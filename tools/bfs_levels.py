#!/usr/bin/env python3
"""Per-level breakdown of the fused BFS: level times from device-side stamps, push kernels per launch slot."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mini_amd
from mini_amd import rmat

ap = argparse.ArgumentParser(); ap.add_argument("--scale", type=int, default=22); ap.add_argument("--runs", type=int, default=3); ap.add_argument("--layout", type=int, default=1); ap.add_argument("--mode", type=int, default=0); ap.add_argument("--alpha", type=float, default=4.0)
a = ap.parse_args()
ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
g = rmat.rmat_csr(ctx, a.scale, 16, seed=a.scale)
graph = mini_amd.Graph.from_device(ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"])
if a.layout:
    graph.build_layout()          # the library's own layout (unit blocks, degree classes, cold-edge lists): what bench.py runs
ro = g["row_offsets"].cpu().numpy()
srcs = rmat.pick_sources(ro, a.runs + 1, a.scale)
bfs = mini_amd.BfsProblem(graph, srcs[0])
bfs.set_kernel_timing(True)
bfs.run(srcs[0], a.mode, a.alpha)
for s in srcs[1:]:
    st = bfs.run(s, a.mode, a.alpha)
    tr, ms, cl = bfs.level_trace(), bfs.level_times_ms(), bfs.level_claims()
    print("src %d: levels %d (small %d) reached %d m_t %d kernel_ms %.3f" % (s, st["levels"], st["small_levels"], st["reached"], st["m_t"], st["kernel_ns"] / 1e6))
    print("  marks %d  push_levels %d  push_edges %d  pull_edges %d" % (st["claims"], st["push_levels"], st["push_edges"], st["pull_edges"]))
    for lv, ((nf, ne), t) in enumerate(zip(tr, ms)):
        b = 8.0 * ne + 20.0 * nf
        print("  level %2d  nf %9d  edges %10d  %8.3f ms  %8.1f GTEPS  %7.1f algGB/s  marks %9d" % (lv, nf, ne, t, ne / t / 1e6 if t > 0 else 0, b / t / 1e6 if t > 0 else 0, cl[lv] if lv < 64 else -1))
    kt = bfs.level_kernel_times_ms()
    print("  slots (stream ms, wave ms): " + "  ".join("(%.3f, %.3f)" % x for x in kt[:8]))
    k = bfs.kernel_times()
    for name in ("stream", "wave"):
        q = k[name]
        print("  %-6s launches %d  %.3f ms  edges %d (%.1f GTEPS)" % (name, q["launches"], q["ns"] / 1e6, q["edges"], q["edges"] / max(q["ns"], 1)))
    ms = bfs.batch_times_ms()
    print("  batches:", ["%.4f" % x for x in ms])
