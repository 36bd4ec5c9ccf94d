"""The library's own multi-rank loops -- mgx_dbfs2_run (include/mgx/bfs_dist2.hpp: d2_run) and mgx_dsssp_run
(include/mgx/sssp_dist.hpp: dsssp_run) -- with 2, 4 and 8 ranks on ONE GPU, over the in-process stand-in for RCCL
(include/mgx/comm_loopback.hpp): G host threads, a stream and an engine each, the same C++ loop, the same buffers, the
same sequence of collectives the eight-GPU job issues (RCCL itself refuses two ranks on one device).

Checked per case: every rank's labels (distances) against the single-GPU traversal of the unpartitioned graph and the CPU
oracle, vertex by vertex; every rank ends with the same visited bitmap; all ranks report the same level count; the edges
the ranks expanded sum to the degrees of the reached vertices.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_gpu(built):
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch


def _contexts(torch, G):
    import mini_amd
    streams = [torch.cuda.Stream() for _ in range(G)]
    return [mini_amd.Context(0, s.cuda_stream) for s in streams], streams


def _bfs_world(torch, scale, G, seed, lists, env=None):
    """G rank engines of the cyclic partition of R-MAT `scale`, a context (stream) each"""
    from mini_amd.dist_bfs import HipRankEngine2, rmat_cyclic_shard
    ctxs, streams = _contexts(torch, G)
    dev = torch.device("cuda", 0)
    old = {}
    env = dict(env or {})
    env["MGX_DIST_LISTS"] = "1" if lists else "0"
    for k, v in env.items():
        old[k] = os.environ.get(k)
        os.environ[k] = v
    try:
        engs, shards = [], []
        for r in range(G):
            with torch.cuda.stream(streams[r]):
                ro, col, new_of_old, old_of_new, deg_new = rmat_cyclic_shard(ctxs[r], scale, 16, seed, G, r, dev)
                engs.append(HipRankEngine2(ctxs[r], 1 << scale, G, r, ro, col))
                shards.append((ro, col))
        torch.cuda.synchronize()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    return ctxs, streams, engs, shards, new_of_old, old_of_new, deg_new


def _single_gpu_labels(gpu_ctx, scale, seed, sources_old):
    import mini_amd
    from mini_amd import rmat
    g = rmat.rmat_csr(gpu_ctx, scale, 16, seed=seed)
    graph = mini_amd.Graph.from_device(gpu_ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"]).build_layout()
    bfs, out = None, {}
    for s in sources_old:
        bfs = bfs or mini_amd.BfsProblem(graph, s)
        bfs.run(s)
        out[s] = bfs.labels().copy()
    return g, out


def _gathered(engs, n):
    G = len(engs)
    lab = np.empty(n, dtype=np.int32)
    for r, e in enumerate(engs):
        lab[r::G] = e.labels()
    return lab


@pytest.mark.parametrize("scale,G,exchange,lists", [
    (16, 2, "gather", True), (16, 2, "reduce", False),
    (17, 4, "reduce", True), (17, 4, "gather", False),
    (18, 8, "reduce", True), (18, 8, "gather", True), (18, 8, "reduce", False),
    (15, 3, "gather", True),                      # not a power of two: the unfused OR-merge + queue build
])
def test_native_bfs_loop_over_loopback_world(gpu_ctx, oracle, torch_gpu, scale, G, exchange, lists):
    torch = torch_gpu
    from mini_amd.dist_bfs import LoopbackComm, run_rank_threads
    ctxs, streams, engs, shards, new_of_old, old_of_new, deg_new = _bfs_world(torch, scale, G, scale, lists)
    n = 1 << scale
    n2o, o2n = new_of_old.cpu().numpy(), old_of_new.cpu().numpy()
    deg = deg_new.cpu().numpy()
    cand = np.nonzero(deg > 0)[0]
    # the biggest hub (a list overflows at level 1), a middling vertex, a leaf (sparse levels at both ends), and a vertex of
    # degree 0 if there is one (a traversal of one level: every rank's frontier but the owner's is empty from the start)
    # Order: the first traversal of an engine looks once per level and leaves the level plan; the leaf's plan (lists at the first
    # levels) then meets the hub, whose neighbourhood overflows a list against it -- the traversal freezes and is continued --, and
    # the hub's plan (bitmaps early) meets the leaf again.
    # (round 6: level 0 of every traversal runs with a host look -- the ranks agree on the level plan there, bfs_dist2.hpp -- so a list
    #  that overflows against the plan must do so at level >= 1 to freeze a traversal: the vertex WITHOUT edges goes first -- its
    #  traversal has no level 1, so the plan made from it sends every later level through the lists -- and the hub behind it freezes)
    iso = np.nonzero(deg == 0)[0]
    srcs = ([int(iso[0])] if len(iso) else []) + [int(cand[0]), int(cand[-1]), int(cand[len(cand) // 2]), int(cand[-1]), int(cand[0])]
    if len(iso):
        srcs.append(int(iso[-1]))
    g, single = _single_gpu_labels(gpu_ctx, scale, scale, sorted({int(o2n[s]) for s in srcs}))
    ro_h, ci_h = g["row_offsets"].cpu().numpy(), g["col_indices"].cpu().numpy()
    ident = LoopbackComm.new_id()
    comms = run_rank_threads(G, lambda r: LoopbackComm(ctxs[r], r, G, ident))
    rounds_before = comms[0].rounds()
    for i, src in enumerate(srcs):
        sts = run_rank_threads(G, lambda r: engs[r].run_native(src, comms[r], exchange))
        assert all(st["over"] for st in sts) and len({st["levels"] for st in sts}) == 1, sts
        lab_new = _gathered(engs, n)
        want = single[int(o2n[src])]
        assert np.array_equal(lab_new[n2o], want), "labels differ from the single-GPU traversal (source %d)" % src
        if i == 0:
            assert np.array_equal(want, oracle.bfs_cpu(ro_h, ci_h, int(o2n[src])))
        reached = lab_new >= 0
        assert int(lab_new.max()) + 1 == sts[0]["levels"]
        assert sum(st["edges_local"] for st in sts) == int(deg[reached].sum())
        want_bits = np.packbits(reached, bitorder="little")
        want_words = np.zeros(engs[0].nwords, dtype=np.uint32)
        want_words.view(np.uint8)[: len(want_bits)] = want_bits
        for e in engs:
            assert np.array_equal(e.visited(), want_words), "rank %d's bitmap differs from the reached set" % e.rank
    assert comms[0].rounds() > rounds_before            # the collectives went through the loopback world
    if lists and G > 1:
        # (round 6) ONE rank loses its history -- a recreated engine handle -- while the others hold a plan: level 0's
        # agreement sends every rank level by level instead of letting them enqueue different collectives (which the loopback world
        # would report as mismatched rounds, and real RCCL would answer with a hang); labels as ever, and nobody planned ahead
        before = [e.spec_stats()[0] for e in engs]
        engs[G - 1].forget_plan()
        src = srcs[1]
        sts = run_rank_threads(G, lambda r: engs[r].run_native(src, comms[r], exchange))
        assert all(st["over"] for st in sts) and len({st["levels"] for st in sts}) == 1, sts
        assert np.array_equal(_gathered(engs, n)[n2o], single[int(o2n[src])]), "labels differ after one rank forgot its plan"
        assert [e.spec_stats()[0] for e in engs] == before, "a rank planned ahead although the ranks' plans differed"
        # ... and the next traversal is planned again on every rank (the histories differ in length, the plans they yield need not)
        sts = run_rank_threads(G, lambda r: engs[r].run_native(src, comms[r], exchange))
        assert np.array_equal(_gathered(engs, n)[n2o], single[int(o2n[src])])
        for e in engs:
            e.forget_plan()                              # (the statistics below count the traversals above this block)
        stats_now = [e.spec_stats() for e in engs]
        assert len({s[0] - b for s, b in zip(stats_now, before)}) == 1, (stats_now, before)     # all planned, or none did
    if lists:
        # sparse levels were merged from id lists AND some level overflowed its list (the hub's neighbourhood): both protocols ran
        paths = [e.path_levels() for e in engs]
        assert all(p[0] >= 1 for p in paths), paths
        # every traversal but the first (and the one from a vertex without edges: over at level 0) was enqueued ahead from the level
        # plan, on every rank alike; the hub behind the vertex without edges froze
        stats = [e.spec_stats() for e in engs]
        assert len(set(stats)) == 1, stats
        extra = (stats_now[0][0] - before[0]) if G > 1 else 0        # (the second traversal of the block above, if it was planned)
        assert stats[0][0] == len(srcs) - 1 - (1 if len(iso) else 0) + extra, stats
        if len(iso):
            assert stats[0][1] >= 1, stats
    for c in comms:
        c.close()
    for e in engs:
        e.close()


def test_loopback_world_reports_a_missing_rank_instead_of_hanging(gpu_ctx, torch_gpu, monkeypatch):
    """a rank that never joins: the others get an error after the deadline (MGX_LOOPBACK_TIMEOUT_S), nobody hangs"""
    import subprocess
    import sys
    code = (
        "import os, sys, time\n"
        "sys.path.insert(0, %r)\n"
        "import torch, mini_amd\n"
        "from mini_amd.dist_bfs import LoopbackComm, run_rank_threads\n"
        "ctxs = [mini_amd.Context(0, torch.cuda.Stream().cuda_stream) for _ in range(2)]\n"
        "ident = LoopbackComm.new_id()\n"
        "t0 = time.time()\n"
        "try:\n"
        "    LoopbackComm(ctxs[0], 0, 2, ident)\n"
        "    print('JOINED')\n"
        "except mini_amd.MgxError as ex:\n"
        "    print('REFUSED after %%.1f s: %%s' %% (time.time() - t0, ex))\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    env = dict(os.environ, MGX_LOOPBACK_TIMEOUT_S="2")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "REFUSED" in r.stdout and "JOINED" not in r.stdout, r.stdout + r.stderr


def test_loopback_collectives_move_the_right_bytes(gpu_ctx, torch_gpu):
    """the stand-in itself against numpy: all-gather (also in place), grouped send / recv of unequal sizes incl. to oneself,
    through the same function table the loops use (mgx_comm_selftest)"""
    torch = torch_gpu
    import ctypes as C
    import mini_amd
    from mini_amd._lib import check, lib
    from mini_amd.dist_bfs import LoopbackComm, run_rank_threads
    G = 4
    ctxs, streams = _contexts(torch, G)
    ident = LoopbackComm.new_id()
    comms = run_rank_threads(G, lambda r: LoopbackComm(ctxs[r], r, G, ident))
    words = 1000
    send = [torch.arange(words * G, dtype=torch.int32, device="cuda") + 100000 * r for r in range(G)]
    gath = [torch.zeros(words * G, dtype=torch.int32, device="cuda") for _ in range(G)]
    a2a = [torch.zeros(words * G, dtype=torch.int32, device="cuda") for _ in range(G)]
    torch.cuda.synchronize()

    def body(r):
        check(lib.mgx_comm_selftest(comms[r]._h, C.c_void_p(send[r].data_ptr()), C.c_void_p(gath[r].data_ptr()),
                                    C.c_void_p(a2a[r].data_ptr()), words))
    run_rank_threads(G, body)
    torch.cuda.synchronize()
    for r in range(G):
        # all-gather of the first `words` words of every rank's send buffer
        want = np.concatenate([send[p][:words].cpu().numpy() for p in range(G)])
        assert np.array_equal(gath[r].cpu().numpy(), want)
        # all-to-all: slice r of every rank's buffer, in rank order
        want = np.concatenate([send[p][r * words:(r + 1) * words].cpu().numpy() for p in range(G)])
        assert np.array_equal(a2a[r].cpu().numpy(), want)
    for c in comms:
        c.close()


@pytest.mark.parametrize("scale,G,lists,spec", [(17, 4, True, "1"), (17, 4, True, "0"), (18, 8, True, "1"), (16, 2, False, "1"), (15, 3, True, "1")])
def test_group_run_in_turn_equals_single_gpu(gpu_ctx, torch_gpu, scale, G, lists, spec):
    """mgx_dbfs2_run_group: all ranks' engines on one context, one host thread, collectives as copies, the level plan of
    mgx_dbfs2_run (the measurement entry of tools/dist2_single.py native): labels equal to the single-GPU traversal"""
    torch = torch_gpu
    from mini_amd.dist_bfs import HipRankEngine2, rmat_cyclic_shard
    dev = torch.device("cuda", 0)
    old = {k: os.environ.get(k) for k in ("MGX_DIST_LISTS", "MGX_DIST_SPEC")}
    os.environ["MGX_DIST_LISTS"] = "1" if lists else "0"
    os.environ["MGX_DIST_SPEC"] = spec
    try:
        engs = []
        for r in range(G):
            ro, col, new_of_old, old_of_new, deg_new = rmat_cyclic_shard(gpu_ctx, scale, 16, scale, G, r, dev)
            engs.append(HipRankEngine2(gpu_ctx, 1 << scale, G, r, ro, col))
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    n = 1 << scale
    n2o, o2n = new_of_old.cpu().numpy(), old_of_new.cpu().numpy()
    deg = deg_new.cpu().numpy()
    cand = np.nonzero(deg > 0)[0]
    iso = np.nonzero(deg == 0)[0]
    srcs = ([int(iso[0])] if len(iso) else []) + [int(cand[0]), int(cand[-1]), int(cand[len(cand) // 3]), int(cand[-2]), int(cand[1])]
    g, single = _single_gpu_labels(gpu_ctx, scale, scale, sorted({int(o2n[s]) for s in srcs}))
    for src in srcs:
        sts = HipRankEngine2.run_group(engs, src)
        assert all(st["over"] for st in sts) and len({st["levels"] for st in sts}) == 1, sts
        lab_new = _gathered(engs, n)
        assert np.array_equal(lab_new[n2o], single[int(o2n[src])]), "labels differ from the single-GPU traversal (source %d)" % src
        assert sum(st["edges_local"] for st in sts) == int(deg[lab_new >= 0].sum())
    if lists:
        stats = engs[0].spec_stats()
        assert stats[0] == (len(srcs) - 1 if spec == "1" else 0), stats
        if spec == "1" and len(iso):
            assert stats[1] >= 1, stats               # the hub behind the vertex without edges (whose plan sends level 1 through the lists)
    for e in engs:
        e.close()


@pytest.mark.parametrize("scale,G", [(12, 2), (13, 4), (14, 8), (12, 3)])
def test_native_sssp_loop_over_loopback_world(gpu_ctx, oracle, torch_gpu, scale, G):
    """mgx_dsssp_run with G ranks: counts all-gather, grouped send / recv of (vertex, distance) pairs, frontier-size all-gather"""
    torch = torch_gpu
    from mini_amd.dist_bfs import LoopbackComm, range_of, run_rank_threads
    from mini_amd.dist_sssp import HipSsspRankEngine
    n, ro, ci, w = oracle.rmat_csr(scale, 16, 40 + scale)
    ctxs, streams = _contexts(torch, G)
    engs = []
    for r in range(G):
        lo, hi = range_of(n, G, r)
        ro_l = (ro[lo:hi + 1] - ro[lo]).astype(np.int32)
        ci_l = ci[ro[lo]:ro[hi]].astype(np.int32)
        w_l = w[ro[lo]:ro[hi]].astype(np.float32)
        with torch.cuda.stream(streams[r]):
            engs.append(HipSsspRankEngine(ctxs[r], n, G, r, torch.from_numpy(ro_l).cuda(), torch.from_numpy(ci_l).cuda(),
                                          torch.from_numpy(w_l).cuda()))
    torch.cuda.synchronize()
    ident = LoopbackComm.new_id()
    comms = run_rank_threads(G, lambda r: LoopbackComm(ctxs[r], r, G, ident))
    deg = np.diff(ro)
    srcs = [int(np.argmax(deg)), int(np.nonzero(deg > 0)[0][-1]), 0]
    for src in srcs:
        sts = run_rank_threads(G, lambda r: engs[r].run_native(src, comms[r]))
        assert len({st["iterations"] for st in sts}) == 1
        got = np.concatenate([e.distances() for e in engs])
        assert np.array_equal(got, oracle.sssp_dijkstra_f32(ro, ci, w, src)), "distances differ from the oracle (source %d)" % src
        assert sum(st["pairs_sent"] for st in sts) == sum(st["pairs_received"] for st in sts)
    for c in comms:
        c.close()
    for e in engs:
        e.close()
