#!/usr/bin/env python3
"""Per-rank throughput of the partitioned-BFS device pieces (world = 1: no exchange)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mini_amd
from mini_amd import rmat
from mini_amd.dist_bfs import HipRankEngine, DistBfs
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
g = rmat.rmat_csr(ctx, scale, 16, seed=scale)
eng = HipRankEngine(ctx, g["n"], 1, 0, g["row_offsets"], g["col_indices"])
bfs = DistBfs(eng, 0, 1, "cuda")
ro = g["row_offsets"].cpu().numpy()
srcs = rmat.pick_sources(ro, 4, scale)
bfs.run(srcs[0])
for s in srcs[1:]:
    torch.cuda.synchronize(); t0 = time.perf_counter(); st = bfs.run(s); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("src %d levels %d edges %d  %.3f ms  %.1f GTEPS" % (s, st["levels"], st["edges_local"], dt * 1e3, st["edges_local"] / dt / 1e9))
