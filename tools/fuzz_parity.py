#!/usr/bin/env python3
"""Randomised parity campaign (not a test: minutes of GPU time): fused BFS (both launch schemes, push and
direction-optimising with random alpha) and the fused SSSP loop against the oracle, on R-MAT graphs of several scales
and on random / structured graphs (uniform random, star forests, grids, long paths with shortcuts).
usage: fuzz_parity.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mini_amd
from tests.oracle_binding import Oracle

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
orc = Oracle()
ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "1")))


def graphs():
    k = 0
    while True:
        k += 1
        if k % 7 == 6:      # big enough for vertices behind the LDS prefix (cold-edge pass): R-MAT 20 / 21, few edges per vertex
            s = int(rng.integers(20, 22))
            n, ro, ci, w = orc.rmat_csr(s, int(rng.integers(2, 7)), int(rng.integers(1, 1 << 30)))
            yield "rmat-%d" % s, n, ro, ci, w
            continue
        kind = k % 5
        if kind == 0:
            s = int(rng.integers(8, 19))
            n, ro, ci, w = orc.rmat_csr(s, int(rng.integers(2, 24)), int(rng.integers(1, 1 << 30)))
            yield "rmat-%d" % s, n, ro, ci, w
            continue
        if kind == 1:       # uniform random, directed or not
            n = int(rng.integers(2, 200000)); e = int(rng.integers(1, 8 * n))
            t0 = rng.integers(0, n, size=e).astype(np.int32); t1 = rng.integers(0, n, size=e).astype(np.int32)
            name = "uniform"
        elif kind == 2:     # star forest: hubs of every size, leaves shared between hubs now and then
            h = int(rng.integers(1, 300)); n = 100000
            deg = rng.integers(1, 3000, size=h)
            t0 = np.repeat(np.arange(h), deg).astype(np.int32); t1 = rng.integers(h, n, size=int(deg.sum())).astype(np.int32)
            name = "stars"
        elif kind == 3:     # grid
            a, b = int(rng.integers(2, 400)), int(rng.integers(2, 400)); n = a * b
            idx = np.arange(n).reshape(a, b)
            t0 = np.concatenate([idx[:, :-1].ravel(), idx[:-1, :].ravel()]).astype(np.int32)
            t1 = np.concatenate([idx[:, 1:].ravel(), idx[1:, :].ravel()]).astype(np.int32)
            name = "grid"
        else:               # long path with random shortcuts
            n = int(rng.integers(100, 3000)); sc = int(rng.integers(0, 20))
            t0 = np.concatenate([np.arange(n - 1), rng.integers(0, n, size=sc)]).astype(np.int32)
            t1 = np.concatenate([np.arange(1, n), rng.integers(0, n, size=sc)]).astype(np.int32)
            name = "path"
        undir = bool(rng.integers(0, 2)) or name in ("grid", "path")
        wv = rng.integers(0, 64, size=len(t0)).astype(np.float32)
        ro, ci, w = orc.csr_from_tuples(n, t0, t1, wv, undir=undir)
        yield name + ("" if undir else "-directed"), n, ro, ci, w


def partitioned_labels(n, ro, ci, src, G, mode):
    """generation-2 partitioned BFS with G rank engines in this process (the collectives as copies): global labels in
    original ids"""
    from mini_amd.dist_bfs import HipRankEngine2, LoopbackComm, cyclic_shard_from_csr, run_rank_threads
    engs, maps = [], None
    # (round 5) "native": the C++ loop itself over all engines in turn (mgx_dbfs2_run_group: level plan, freeze, continuation);
    # "loopback": mgx_dbfs2_run on G rank THREADS over the in-process communicator, an engine and a stream each
    ctxs = [mini_amd.Context(0, torch.cuda.Stream().cuda_stream) for _ in range(G)] if mode == "loopback" else [ctx] * G
    for r in range(G):
        ro_l, ci_l, new_of_old, old_of_new = cyclic_shard_from_csr(ro, ci, G, r)
        engs.append(HipRankEngine2(ctxs[r], n, G, r, torch.from_numpy(ro_l).cuda(), torch.from_numpy(ci_l).cuda()))
    torch.cuda.synchronize()
    s_new = int(new_of_old[src])
    if mode in ("native", "loopback"):
        d = np.diff(ro)
        # a leaf first (its plan expects sparse early levels), then the biggest hub (which overflows a list against that plan: a
        # frozen traversal), then the source that is checked -- under the plan the two before it left
        warm = [int(np.argmin(np.where(d > 0, d, 1 << 30))), int(np.argmax(d))]
        comms = None
        if mode == "loopback":
            ident = LoopbackComm.new_id()
            comms = run_rank_threads(G, lambda r: LoopbackComm(ctxs[r], r, G, ident))
        exch = str(rng.choice(["gather", "reduce"]))
        for s_old in warm + [src]:
            s2 = int(new_of_old[s_old])
            if mode == "native":
                sts = HipRankEngine2.run_group(engs, s2)
            else:
                sts = run_rank_threads(G, lambda r: engs[r].run_native(s2, comms[r], exch))
            assert all(st["over"] for st in sts) and len({st["levels"] for st in sts}) == 1, sts
        if comms:
            for c in comms:
                c.close()
        lab_new = np.empty(n, dtype=np.int32)
        for r, e in enumerate(engs):
            lab_new[r::G] = e.labels()
            e.close()
        out = np.empty(n, dtype=np.int32)
        out[old_of_new] = lab_new
        return out
    for e in engs:
        e.reset(s_new)
    level = 0
    while mode == "lists":
        # the level protocol of mgx_dbfs2_run: id lists first, the bitmaps only when some rank's list overflowed
        bits = [e.push(level) for e in engs]
        glists = torch.cat([e.list for e in engs]) if G > 1 else engs[0].list
        res = [e.apply_lists(level, glists, G) for e in engs]
        assert len(set(res)) == 1
        level += 1
        if res[0][1] == 0:
            break
        if res[0][0]:
            gathered = torch.cat(bits) if G > 1 else bits[0]
            for e in engs:
                e.merge(level - 1, gathered, G)
    if mode == "lists":
        assert engs[0].status(level)["over"]
    while mode != "lists":
        for _ in range(3):
            bits = [e.push(level) for e in engs]
            if mode == "reduce" and G > 1:
                S = bits[0].numel() // G
                merged = []
                for r, e in enumerate(engs):
                    recv = torch.cat([m[r * S:(r + 1) * S] for m in bits])
                    e.or_maps(recv, G, recv[:S])
                    merged.append(recv[:S])
                full = torch.cat(merged)
                for e in engs:
                    e.merge(level, full, 1)
            else:
                gathered = torch.cat(bits)
                for e in engs:
                    e.merge(level, gathered, G)
            level += 1
        if engs[0].status(level)["over"]:
            break
    lab_new = np.empty(n, dtype=np.int32)
    for r, e in enumerate(engs):
        lab_new[r::G] = e.labels()
        e.close()
    out = np.empty(n, dtype=np.int32)
    out[old_of_new] = lab_new
    return out


t_end = time.time() + budget
ran = 0
for name, n, ro, ci, w in graphs():
    if time.time() > t_end:
        break
    deg = np.diff(ro)
    d_ro, d_ci, d_w = torch.from_numpy(ro).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(w).cuda()
    g = mini_amd.Graph.from_device(ctx, n, len(ci), d_ro, d_ci, d_w)
    symmetric = "directed" not in name
    layout = bool(rng.integers(0, 2)) or n > 600000
    if layout:
        # (the short rows' cold-edge lists are built by default only above 2^23 vertices: here by the draw, on the graphs that have cold entries)
        if n >= (1 << 20):
            os.environ["MGX_BFS_COLD_LISTS"] = str(rng.choice(["1", "2"]))
        g.build_layout(weights=True)
        os.environ.pop("MGX_BFS_COLD_LISTS", None)
    bfs, sssp = mini_amd.BfsProblem(g, 0), mini_amd.SsspProblem(g, 0)
    srcs = [int(np.argmax(deg))] + [int(x) for x in rng.integers(0, n, size=4)]
    for src in srcs:
        want = orc.bfs_cpu(ro, ci, src)
        # the traversal's per-level choices forced both ways: chains of small levels (cap), unit blocks (never /
        # whenever the frontier bitmap allows)
        for direct in ("1", "0"):
            os.environ["MGX_BFS_CHAIN_MAX_EDGES"] = str(int(rng.choice([0, 1, 64, 6144]))) if direct == "0" else "6144"
            os.environ["MGX_BFS_DENSE"] = "1000000" if direct == "0" else str(int(rng.choice([0, 2, 16])))
            # ... and this round's: lazy queues, vertex-by-vertex short rows, deferred marks, in-place chain launches, cold pass
            knobs = {"MGX_BFS_LAZY": rng.choice(["", "0", "4", "1048576"]), "MGX_BFS_VSHORT": rng.choice(["", "0", "1000000"]),
                     "MGX_BFS_DEFER": rng.choice(["", "0", "1", "2048"]), "MGX_BFS_SEED_CHAIN": rng.choice(["", "0"]),
                     "MGX_BFS_TAIL_CHAIN": rng.choice(["", "0"]), "MGX_BFS_CHAIN_BIG_EDGES": rng.choice(["", "100", "12288"]),
                     "MGX_BFS_COLD": rng.choice(["", "0", "2", "1"]), "MGX_BFS_DEFER_REACH": rng.choice(["", "0/1", "4/1"]),
                     "MGX_BFS_MERGED_PULL": rng.choice(["", "0"]), "MGX_BFS_DO_CHAIN": rng.choice(["", "0"]),
                     "MGX_SSSP_BUILD_LIST": rng.choice(["", "1"]), "MGX_BFS_MINI": rng.choice(["", "0", "2"]),
                     "MGX_SSSP_DENSE": rng.choice(["", "0", "1000000000"]),
                     # round 4: 24-bit unit blocks on / off, the deferred range, the per-source launch plan
                     "MGX_BFS_PACK24": rng.choice(["", "0"]), "MGX_BFS_DEFER_WORDS": rng.choice(["", "32", "4096", "0"]),
                     "MGX_BFS_SRC_PLAN": rng.choice(["", "0"])}
            for kk, vv in knobs.items():
                if vv == "":
                    os.environ.pop(kk, None)
                else:
                    os.environ[kk] = str(vv)
            st = bfs.run(src)
            assert np.array_equal(bfs.labels(), want), (name, n, src, "push", direct, layout, knobs)
            assert st["m_t"] == int(deg[want >= 0].sum()), (name, n, src, "m_t", direct, layout)
            if symmetric:
                alpha = float(10.0 ** rng.uniform(-2, 4))
                bfs.run(src, mode=mini_amd.MGX_BFS_DIRECTION_OPT, alpha=alpha)
                assert np.array_equal(bfs.labels(), want), (name, n, src, "do", alpha, direct, layout)
        for kk in ("MGX_BFS_CHAIN_MAX_EDGES", "MGX_BFS_DENSE", "MGX_BFS_LAZY", "MGX_BFS_VSHORT", "MGX_BFS_DEFER", "MGX_BFS_SEED_CHAIN",
                   "MGX_BFS_TAIL_CHAIN", "MGX_BFS_CHAIN_BIG_EDGES", "MGX_BFS_COLD", "MGX_BFS_DEFER_REACH", "MGX_BFS_MERGED_PULL",
                   "MGX_BFS_DO_CHAIN", "MGX_BFS_MINI", "MGX_BFS_PACK24", "MGX_BFS_DEFER_WORDS", "MGX_BFS_SRC_PLAN"):
            os.environ.pop(kk, None)
        if src == srcs[0] and (n <= 150000 or (n >= (1 << 20) and ran % 2 == 0)):
            # the partitioned engine's rank engines in this process: small graphs, and R-MAT 20 / 21 (the ranks' cold-edge pass);
            # unit blocks forced onto every eligible level or by the default rule
            G = int(rng.choice([2, 3, 5, 8])) if n <= 150000 else 2
            mode = str(rng.choice(["gather", "reduce", "lists", "native", "native", "loopback"]))
            dd = str(rng.choice(["", "1000000"]))
            if dd:
                os.environ["MGX_DIST_DENSE_DIV"] = dd
            os.environ["MGX_DIST_COLD"] = str(rng.choice(["1", "1", "0"]))
            # round 4's rank-engine paths, each against its switch
            dknobs = {"MGX_DIST_SPARSE_PUSH": rng.choice(["", "0"]), "MGX_DIST_DECLARE_MUL": rng.choice(["", "0", "1"]),
                      "MGX_DIST_FUSED_MERGE": rng.choice(["", "0"]), "MGX_DIST_COLD_REDUCE": rng.choice(["", "0"]),
                      "MGX_DIST_HOT_UNITS": rng.choice(["", "0"]), "MGX_BFS_COLD_PACK": rng.choice(["", "0"]),
                      "MGX_DIST_DEFER": rng.choice(["", "0", "2", "2"]), "MGX_DIST_VSHORT": rng.choice(["", "0", "1000000"]),
                      "MGX_DIST_COLD_WGS": rng.choice(["", "3", "700"])}
            for kk, vv in dknobs.items():
                if vv == "":
                    os.environ.pop(kk, None)
                else:
                    os.environ[kk] = str(vv)
            got = partitioned_labels(n, ro, ci, src, G, mode)
            os.environ.pop("MGX_DIST_DENSE_DIV", None); os.environ.pop("MGX_DIST_COLD", None)
            for kk in dknobs:
                os.environ.pop(kk, None)
            assert np.array_equal(got, want), (name, n, src, "partitioned", G, mode, dd, dknobs)
        dist, _, _ = orc.sssp_enact(ro, ci, w, src, 8.0)
        sssp.run(src)                                  # (MGX_SSSP_BUILD_LIST: whatever the last draw left)
        assert np.array_equal(sssp.distances(), dist), (name, n, src, "sssp", layout, os.environ.get("MGX_SSSP_BUILD_LIST"))
        os.environ.pop("MGX_SSSP_BUILD_LIST", None)
    # a batch of sources (mgx_bfs_run_many): the counters of every traversal, the labels of the last one
    sts, reruns = bfs.run_many(srcs)
    for src, st in zip(srcs, sts):
        w_ = orc.bfs_cpu(ro, ci, src)
        assert st["m_t"] == int(deg[w_ >= 0].sum()) and st["reached"] == int((w_ >= 0).sum()), (name, n, src, "run_many", st, reruns)
    assert np.array_equal(bfs.labels(), orc.bfs_cpu(ro, ci, srcs[-1])), (name, n, "run_many labels", reruns)
    # the neighbour-reduce over the full frontier (mgx/nreduce.hpp on graphs with a layout, the general kernel otherwise)
    if len(ci) > 0:
        ids = np.arange(n, dtype=np.int32)
        f = mini_amd.Frontier(ctx, n).load(ids)
        vals = rng.integers(-1000, 1000, size=n).astype(np.int32)
        red = torch.full((n,), 12345, dtype=torch.int32, device="cuda")
        nz = mini_amd.segreduce(g, f, torch.from_numpy(vals).cuda(), 2**31 - 1, red, "i32_min")
        want_r, wnz = orc.neighbor_reduce_i32(ro, ci, ids, vals, 2**31 - 1, False)
        assert nz == wnz and np.array_equal(red.cpu().numpy(), want_r), (name, n, "neighbour-reduce", layout)
        fv = rng.integers(0, 8, size=n).astype(np.float32)
        redf = torch.full((n,), -1, dtype=torch.float32, device="cuda")
        mini_amd.segreduce(g, f, torch.from_numpy(fv).cuda(), 0.0, redf, "f32_plus")
        want_f, _ = orc.neighbor_reduce_f32_plus(ro, ci, ids, fv, 0.0)
        assert np.array_equal(redf.cpu().numpy(), want_f), (name, n, "neighbour-reduce f32", layout)
    # k-core decomposition (round 4): the enactor on filter / advance<has_output=false> / filter against the validator's restatement
    if symmetric and n <= 300000 and len(ci) > 0 and ran % 3 == 0:
        kc = mini_amd.KcoreProblem(g)
        largest, kst = kc.enact()
        wc, wl = orc.kcore_cpu(ro, ci)
        assert largest == wl and np.array_equal(kc.num_cores(), wc), (name, n, "k-core", largest, wl)
        kc.close()
    ran += 1
    print("ok %-18s n=%-7d m=%-9d layout=%d" % (name, n, len(ci), layout), flush=True)
print("fuzz: %d graphs, all equal to the oracle" % ran)
