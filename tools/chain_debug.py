#!/usr/bin/env python3
"""level structure and slot use of a few traversals (which levels the in-place chain launches took)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mini_amd
from mini_amd import rmat
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
g = rmat.rmat_csr(ctx, scale, 16, seed=scale)
graph = mini_amd.Graph.from_device(ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"]).build_layout()
ro = g["row_offsets"].cpu().numpy()
srcs = rmat.pick_sources(ro, 8, scale)
bfs = mini_amd.BfsProblem(graph, srcs[0])
for rep in range(2):
    for s in srcs:
        st = bfs.run(s)
        print("src %8d levels %d slots %d small %d launches %d  trace %s" % (s, st["levels"], st["slots"], st["small_levels"], st["kernel_launches"],
              " ".join("%d/%d" % t for t in bfs.level_trace())), flush=True)
