#!/usr/bin/env python3
"""The library's own partitioned-BFS loop (mgx_dbfs2_run) with G ranks on ONE GPU over the loopback communicator
(include/mgx/comm_loopback.hpp): G host threads, an engine and a stream each, the collectives as device copies.
What tools/dist2_single.py measures with the Python superstep loop, here with the C++ loop the 8-GPU job runs --
host looks, launch gaps and all.  The G ranks share the device, so wall time / G is what one rank's GPU spends per
traversal if the other ranks' kernels fill the device while it waits (an upper bound of the kernel time per rank,
no xGMI time in it).
usage: dist2_loopback.py [scale] [G] [gather|reduce] [sources]   (DIST2_CHECK=1: labels against the single-GPU run)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import mini_amd
from mini_amd.dist_bfs import HipRankEngine2, LoopbackComm, rmat_cyclic_shard, run_rank_threads

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
exchange = sys.argv[3] if len(sys.argv) > 3 else "reduce"
nsrc = int(sys.argv[4]) if len(sys.argv) > 4 else 8
dev = torch.device("cuda", 0)
n = 1 << scale
streams = [torch.cuda.Stream() for _ in range(G)]
ctxs = [mini_amd.Context(0, s.cuda_stream) for s in streams]
engs = []
for r in range(G):
    with torch.cuda.stream(streams[r]):
        ro, col, new_of_old, old_of_new, deg_new = rmat_cyclic_shard(ctxs[r], scale, 16, scale, G, r, dev)
        engs.append(HipRankEngine2(ctxs[r], n, G, r, ro, col))
torch.cuda.synchronize()
ident = LoopbackComm.new_id()
comms = run_rank_threads(G, lambda r: LoopbackComm(ctxs[r], r, G, ident))
cand = torch.nonzero(deg_new > 0)[:, 0]
srcs = [int(cand[(i * 7919 + 1) * len(cand) // (nsrc * 7919 + 7)]) for i in range(nsrc)]
times = []
for it, s in enumerate([srcs[0]] + srcs):                     # (the first run warms the engines' level hints)
    torch.cuda.synchronize()
    r0 = comms[0].rounds()
    t0 = time.perf_counter()
    sts = run_rank_threads(G, lambda r: engs[r].run_native(s, comms[r], exchange))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rounds = comms[0].rounds() - r0
    edges = sum(st["edges_local"] for st in sts)
    if it:
        times.append(dt)
        print("src %d levels %d edges %d  wall %.3f ms  per-rank %.3f ms  -> %.1f GTEPS aggregate (loopback: no xGMI time)  collective rounds %d  path %r"
              % (s, sts[0]["levels"], edges, dt * 1e3, dt * 1e3 / G, edges / (dt / G) / 1e9, rounds, engs[0].path_levels()))
    if it == 1 and os.environ.get("DIST2_CHECK") == "1":
        from mini_amd import rmat
        g = rmat.rmat_csr(ctxs[0], scale, 16, seed=scale)
        graph = mini_amd.Graph.from_device(ctxs[0], g["n"], g["m"], g["row_offsets"], g["col_indices"]).build_layout()
        src_old = int(old_of_new[s])
        bfs = mini_amd.BfsProblem(graph, src_old)
        bfs.run(src_old)
        lab_new = np.empty(n, dtype=np.int32)
        for r, e in enumerate(engs):
            lab_new[r::G] = e.labels()
        same = bool(np.array_equal(lab_new[new_of_old.cpu().numpy()], bfs.labels()))
        print("check vs the single-GPU traversal of the unpartitioned graph: labels equal = %s" % same)
        assert same
print("median per-rank %.3f ms over %d sources (scale %d, %d ranks, exchange %s)" % (float(np.median(times)) * 1e3 / G, len(times), scale, G, exchange))
