#!/usr/bin/env python3
"""BFS on the operator-per-superstep path (the reference's enact_pushpull loop, bfs_enactor.hxx:41-117, on our
advance / filter kernels) next to the fused device-resident traversal, same graph and sources."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mini_amd
from mini_amd import rmat
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
g = rmat.rmat_csr(ctx, scale, 16, seed=scale)
graph = mini_amd.Graph.from_device(ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"])
graph.attach_layout(*rmat.degree_order(g["row_offsets"], g["col_indices"]))
ro = g["row_offsets"].cpu().numpy()
deg = np.diff(ro)
srcs = rmat.pick_sources(ro, 5, scale)
bfs = mini_amd.BfsProblem(graph, srcs[0])
bfs.reset(srcs[0]); bfs.enact_pushpull(); ctx.synchronize()
for alpha, name in ((None, "push only (alpha = 1/n)"), (4.0, "push/pull alpha = 4")):
    tot_t, tot_e = 0.0, 0
    for s in srcs[1:]:
        bfs.reset(s); ctx.synchronize()
        t0 = time.perf_counter(); bfs.enact_pushpull(alpha); ctx.synchronize(); dt = time.perf_counter() - t0
        lab = bfs.labels(); tot_t += dt; tot_e += int(deg[lab >= 0].sum())
    print("operator path, %s: %.3f ms per traversal, %.1f GTEPS" % (name, tot_t / 4 * 1e3, tot_e / tot_t / 1e9))
# the reference's idempotent mode: advance emits every neighbour (no label test, no CAS), uniquify culls + labels
bfs.reset(srcs[0]); bfs.enact_idempotent(); ctx.synchronize()
tot_t, tot_e = 0.0, 0
for s in srcs[1:]:
    bfs.reset(s); ctx.synchronize()
    t0 = time.perf_counter(); bfs.enact_idempotent(); ctx.synchronize(); dt = time.perf_counter() - t0
    lab = bfs.labels(); tot_t += dt; tot_e += int(deg[lab >= 0].sum())
print("operator path, idempotent advance + uniquify: %.3f ms per traversal, %.1f GTEPS" % (tot_t / 4 * 1e3, tot_e / tot_t / 1e9))
tot_t, tot_e = 0.0, 0
for s in srcs[1:]:
    ctx.synchronize(); t0 = time.perf_counter(); st = bfs.run(s); ctx.synchronize(); dt = time.perf_counter() - t0
    tot_t += dt; tot_e += st["m_t"]
print("fused path: %.3f ms per traversal, %.1f GTEPS" % (tot_t / 4 * 1e3, tot_e / tot_t / 1e9))
