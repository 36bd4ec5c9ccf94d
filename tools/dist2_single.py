#!/usr/bin/env python3
"""Generation-2 partitioned BFS on ONE GPU: G rank engines share the device and run one after another
(the all-gather becomes a concatenation), so the time per rank = total / G is what each GPU of a G-GPU
job would spend in kernels per BFS (no xGMI time).  usage: dist2_single.py [scale] [G] [native|lists|gather|reduce]
native: the C++ loop itself (mgx_dbfs2_run_group: the engines in turn from one host thread, the collectives as device copies, the
level plan of mgx_dbfs2_run -- a whole traversal enqueued ahead, one host wait): wall time / G = kernels and launch gaps of one
rank per traversal, no interpreter between the launches;
lists (default): the level protocol of mgx_dbfs2_run / DistBfs2.run -- id lists first (the all-gather a concatenation, every
engine's list merge with its host round trip), the bitmaps only when some rank's list overflowed;
gather: bitmaps on every level; reduce: the slice exchange of DistBfs2 (all-to-all of slices -> OR -> all-gather of merged
slices), the two collectives again as copies."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mini_amd
from mini_amd.dist_bfs import HipRankEngine2, rmat_cyclic_shard
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
G = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mode = sys.argv[3] if len(sys.argv) > 3 else "lists"
dev = torch.device("cuda", 0)
ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
n = 1 << scale
engs = []
t_shard = t_eng = 0.0
for r in range(G):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ro, col, new_of_old, old_of_new, deg_new = rmat_cyclic_shard(ctx, scale, 16, scale, G, r, dev)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    engs.append(HipRankEngine2(ctx, n, G, r, ro, col))
    torch.cuda.synchronize(); t_shard += t1 - t0; t_eng += time.perf_counter() - t1
# what ONE rank spends before its first traversal: its shard out of the whole pair stream (every rank hashes all edgefactor * n
# pairs twice -- degrees, then its own rows' keys -- and sorts its keys), then unit blocks + cold-edge lists of its rows
print("setup per rank: shard (mgx_dbfs2_shard_plan + _fill) %.3f s, engine (unit blocks, cold-edge lists) %.3f s" % (t_shard / G, t_eng / G))
srcs = [int(v) for v in torch.nonzero(deg_new > 0)[:: max(1, n // 64)][:6, 0].tolist()]
hint = 8
for it, s in enumerate(srcs):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if mode != "native":
        for e in engs:
            e.reset(s)
    level, batch = 0, hint
    sparse = dense = 0
    if mode == "native":
        sts = HipRankEngine2.run_group(engs, s)
        level = sts[0]["levels"]
    while mode == "lists":
        maps = [e.push(level) for e in engs]
        glists = torch.cat([e.list for e in engs]) if G > 1 else engs[0].list
        res = [e.apply_lists(level, glists, G) for e in engs]
        level += 1
        if res[0][1] == 0:
            break
        if res[0][0]:
            gathered = torch.cat(maps) if G > 1 else maps[0]
            for e in engs:
                e.merge(level - 1, gathered, G)
            dense += 1
        else:
            sparse += 1
    while mode not in ("lists", "native"):
        for _ in range(batch):
            if mode == "reduce" and G > 1:
                maps = [e.push(level) for e in engs]
                S = maps[0].numel() // G
                merged = []
                for r, e in enumerate(engs):                       # rank r: slice r of every map, OR-ed
                    recv = torch.cat([m[r * S:(r + 1) * S] for m in maps])
                    e.or_maps(recv, G, recv[:S])
                    merged.append(recv[:S])
                full = torch.cat(merged)
                for e in engs:
                    e.merge(level, full, 1)
            else:
                gathered = torch.cat([e.push(level) for e in engs]) if G > 1 else engs[0].push(level)
                for e in engs:
                    e.merge(level, gathered, G)
            level += 1
        sts = [e.status(level) for e in engs]
        if sts[0]["over"]:
            break
        batch = 2
    if mode == "lists":
        sts = [e.status(level) for e in engs]
    hint = sts[0]["levels"] + 1
    edges = sum(st["edges_local"] for st in sts)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if it:
        if os.environ.get("DIST2_CHECK") == "1" and it == 1:
            # the same traversal on the unpartitioned graph (fused single-GPU path), labels compared vertex by vertex
            from mini_amd import rmat
            g = rmat.rmat_csr(ctx, scale, 16, seed=scale)
            graph = mini_amd.Graph.from_device(ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"]).build_layout()
            n2o = new_of_old.long()
            src_old = int(torch.nonzero(n2o == s)[0, 0])
            bfs = mini_amd.BfsProblem(graph, src_old)
            bfs.run(src_old)
            single = torch.from_numpy(bfs.labels())
            lab_new = torch.empty(n, dtype=torch.int32)
            for r, e in enumerate(engs):
                lab_new[r::G] = torch.from_numpy(e.labels())
            same = bool(torch.equal(lab_new[n2o.cpu()], single))
            print("check vs the single-GPU traversal of the unpartitioned graph: labels equal = %s (reached %d)" % (same, int((single >= 0).sum())))
            assert same
        print(mode + " src %d levels %d edges %d  total %.3f ms  per-rank %.3f ms  -> %.1f GTEPS aggregate (no exchange time); levels read from unit blocks on rank 0: %d"
              % (s, sts[0]["levels"], edges, dt * 1e3, dt * 1e3 / G, edges / (dt / G) / 1e9, engs[0].dense_levels())
              + (" (levels merged from lists %d, from bitmaps %d)" % (sparse, dense) if mode == "lists" else "")
              + (" (plan: %r)" % (engs[0].spec_stats(),) if mode == "native" else ""))
