#!/usr/bin/env python3
"""per-level picture of direction-optimising traversals: sizes, stamps, what ran where.  usage: do_levels.py [scale] [alpha] [sources]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mini_amd
from mini_amd import rmat
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
alpha = float(sys.argv[2]) if len(sys.argv) > 2 else 64.0
nsrc = int(sys.argv[3]) if len(sys.argv) > 3 else 4
ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
g = rmat.rmat_csr(ctx, scale, 16, seed=scale)
graph = mini_amd.Graph.from_device(ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"]).build_layout()
ro = g["row_offsets"].cpu().numpy()
srcs = rmat.pick_sources(ro, nsrc + 2, scale)
bfs = mini_amd.BfsProblem(graph, srcs[0])
for s in srcs[:2]:
    bfs.run(s, mode=mini_amd.MGX_BFS_DIRECTION_OPT, alpha=alpha)
for s in srcs[2:]:
    st = bfs.run(s, mode=mini_amd.MGX_BFS_DIRECTION_OPT, alpha=alpha)
    lv = st["levels"]
    tr = bfs.level_trace(lv)
    us = [x * 1e3 for x in bfs.level_times_ms(min(lv, 63))]
    print("src %d: levels %d slots %d push_levels %d small %d mini %d dense %d lazy %d  device %.1f us" % (
        s, lv, st["slots"], st["push_levels"], st["small_levels"], st.get("mini_slots", -1), st["dense_slots"], st["lazy_slots"], sum(us)))
    print("   sizes   " + "  ".join("%d/%d" % (v, e) for v, e in tr))
    print("   us      " + "  ".join("%.1f" % x for x in us))
