"""BASELINE.json configs at their NAMED shapes, through the C-ABI on the GPU:

  config 2  BFS push on RMAT-22 ef 16, symmetrised        -> full size, one source against the oracle
  config 3  SSSP on the same topology, integer weights      -> full size, one source against the oracle's float Dijkstra
  config 4  direction-optimal BFS on a DIRECTED graph with a genuine CSC built by the library (the stand-in SURVEY 8d
            names for soc-LiveJournal: R-MAT without symmetrisation), alpha in {1/n, .01, .05, .1, 1}: oracle at scale
            16, size-independent BFS-tree properties at scale 20 and 22; the real file through mgx_load_mtx when
            MGX_DATA_DIR holds it
  config 5  BFS on RMAT-26 ef 16 partitioned over 8 ranks: the 8 shards built by the library (mgx_dbfs2_shard_*), the 8
            rank engines run in turn on the ONE GPU of the test box (the exchange a concatenation): BFS-tree properties on
            every rank's rows, every rank's bitmap equal; RMAT-23 over 4 ranks the same way with labels equal to the
            single-GPU traversal and the oracle.  The message passing itself: tests/test_dist.py (gloo worlds).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ALPHAS = ("1/n", 0.01, 0.05, 0.1, 1.0)      # SURVEY 8d's list
ALPHAS_PULL = ALPHAS + (64.0, 1e9)           # ... plus two that do reach the bottom-up phase on a directed graph (many vertices unreachable)


def _alpha(a, n):
    return 1.0 / n if a == "1/n" else float(a)


def _directed_rmat_host(oracle, scale, seed):
    """(n, ro, ci, w, co, ri, cw): directed R-MAT CSR (pair (u, v): row v, neighbour u -- graph.hxx:139-157) and its
    transpose, both by the oracle's loader restatement"""
    n = 1 << scale
    s, d, w = oracle.rmat_edges(scale, 0, 16 * n, seed, True)
    ro, ci, ww = oracle.csr_from_tuples(n, s, d, w, undir=False)
    co, ri, cw = oracle.csr_from_tuples(n, d, s, w, undir=False)
    return n, ro, ci, ww, co, ri, cw


@pytest.mark.parametrize("scale", [8, 13])
def test_library_built_csc_is_the_transpose(gpu_ctx, oracle, scale):
    """mgx_graph_build_csc (device radix sort of the edges by destination) against the oracle's transpose: offsets,
    sources of every in-edge list (ascending, duplicates kept) and the weights that travel with them"""
    import mini_amd
    n, ro, ci, w, co, ri, cw = _directed_rmat_host(oracle, scale, 7 + scale)
    g = mini_amd.Graph.from_host(gpu_ctx, ro, ci, w)
    a0 = g.csc_arrays()
    assert np.array_equal(a0[0], ro) and np.array_equal(a0[1], ci)          # before: the CSC slots alias the CSR (F8)
    g.build_csc()
    got_co, got_ri, got_w = g.csc_arrays()
    assert np.array_equal(got_co, co)
    assert np.array_equal(got_ri, ri)
    # equal (column, source) pairs may carry their weights in either order: compare per pair as multisets
    key = np.repeat(np.arange(n, dtype=np.int64), np.diff(co)) * n + ri
    o1, o2 = np.lexsort((got_w, key)), np.lexsort((cw, key))
    assert np.array_equal(got_w[o1], cw[o2])


def test_loader_output_through_the_library_csc(gpu_ctx):
    """the reference's own directed fixture: product loader (mgx_load_mtx) -> upload -> mgx_graph_build_csc, against the
    loader's own _genuine_csc transpose (mgx_load_mtx_csc, host side: tests/test_capi_boundary.py pins that one)"""
    import mini_amd
    path = os.path.join(os.path.dirname(__file__), "golden", "sssp_test.mtx")
    n, ro, ci, w, co, ri, cw = mini_amd.load_mtx(path, undir=False, genuine_csc=True)
    g = mini_amd.Graph.from_host(gpu_ctx, ro, ci, w).build_csc()
    got_co, got_ri, got_w = g.csc_arrays()
    assert np.array_equal(got_co, co) and np.array_equal(got_ri, ri) and np.array_equal(got_w, cw)


def test_config4_directed_rmat16_direction_optimal_vs_oracle(gpu_ctx, oracle):
    """config 4 stand-in at a size the oracle finishes in seconds: directed R-MAT-16, genuine CSC built by the library;
    fused direction-optimising run and the reference's enact_pushpull loop on the operators, alpha swept over
    SURVEY 8d's list -- labels equal the top-down oracle for every switch point"""
    import mini_amd
    scale = 16
    n, ro, ci, w, co, ri, cw = _directed_rmat_host(oracle, scale, 16)
    g = mini_amd.Graph.from_host(gpu_ctx, ro, ci, w).build_csc()
    deg = np.diff(ro)
    srcs = [int(np.argmax(deg))] + [int(v) for v in np.where(deg > 0)[0][[5, 1000]]]
    bfs = mini_amd.BfsProblem(g, srcs[0])
    pulled = 0
    for src in srcs:
        want = oracle.bfs_cpu(ro, ci, src)
        st = bfs.run(src)
        assert np.array_equal(bfs.labels(), want), ("push", src)
        for a in ALPHAS_PULL:
            alpha = _alpha(a, n)
            st = bfs.run(src, mode=mini_amd.MGX_BFS_DIRECTION_OPT, alpha=alpha)
            assert np.array_equal(bfs.labels(), want), ("fused do", src, a)
            assert st["reached"] == int((want >= 0).sum()) and st["m_t"] == int(deg[want >= 0].sum())
            pulled += st["pull_edges"]
            bfs.reset(src)
            bfs.enact_pushpull(alpha)
            assert np.array_equal(bfs.labels(), want), ("enact_pushpull", src, a)
            rc, olabels, _ = oracle.bfs_enact_pushpull(ro, ci, src, alpha, co, ri)
            assert np.array_equal(olabels, want), ("oracle loop", src, a)
    assert pulled > 0             # some alpha of the list did switch to bottom-up


def _bfs_tree_properties(torch, ro, ci, co, ri, labels, src):
    """Size-independent check of a label array on the device (all tensors int32/int64 on cuda): source 0; every edge
    row -> neighbour goes at most one level down; every reached vertex other than the source has an in-neighbour one
    level up; unreached vertices have no reached in-neighbour."""
    n = labels.numel()
    lab = labels.to(torch.int64)
    assert int(lab[src]) == 0
    rows = torch.repeat_interleave(torch.arange(n, device=lab.device), (ro[1:] - ro[:-1]).to(torch.int64))
    lr, lc = lab[rows], lab[ci.to(torch.int64)]
    reached_r = lr >= 0
    assert bool(((lc[reached_r] >= 0) & (lc[reached_r] <= lr[reached_r] + 1)).all()), "an edge skips a level"
    del rows, lr, lc, reached_r
    cols = torch.repeat_interleave(torch.arange(n, device=lab.device), (co[1:] - co[:-1]).to(torch.int64))
    lin = lab[ri.to(torch.int64)]
    big = torch.iinfo(torch.int64).max
    lin = torch.where(lin >= 0, lin, torch.full_like(lin, big))
    best = torch.full((n,), big, dtype=torch.int64, device=lab.device)
    best.scatter_reduce_(0, cols, lin, reduce="amin", include_self=True)
    reached = lab >= 0
    not_src = torch.ones(n, dtype=torch.bool, device=lab.device)
    not_src[src] = False
    assert bool((best[reached & not_src] == lab[reached & not_src] - 1).all()), "a vertex without a parent one level up"
    assert bool((best[~reached] == big).all()), "an unreached vertex with a reached in-neighbour"


@pytest.mark.parametrize("scale", [20, 22])
def test_config4_directed_rmat_full_size_properties(gpu_ctx, torch_mod, scale):
    """config 4 stand-in at scale 20 and at the named scale 22: directed R-MAT built on the device, genuine CSC by the
    library; push-only and direction-optimising runs over the alpha list give ONE label array, and that array has the
    BFS-tree properties"""
    import mini_amd
    from mini_amd import rmat
    torch = torch_mod
    g = rmat.rmat_csr(gpu_ctx, scale, 16, seed=scale, undirected=False)
    n = g["n"]
    graph = mini_amd.Graph.from_device(gpu_ctx, n, g["m"], g["row_offsets"], g["col_indices"]).build_csc()
    co_h, ri_h, _ = graph.csc_arrays()
    co, ri = torch.from_numpy(co_h).cuda(), torch.from_numpy(ri_h).cuda()
    ro_h = g["row_offsets"].cpu().numpy()
    src = rmat.pick_sources(ro_h, 1, scale)[0]
    bfs = mini_amd.BfsProblem(graph, src)
    st0 = bfs.run(src)
    ref = torch.from_numpy(bfs.labels()).cuda()
    _bfs_tree_properties(torch, g["row_offsets"], g["col_indices"], co, ri, ref, src)
    pulled = 0
    for a in ALPHAS_PULL:
        st = bfs.run(src, mode=mini_amd.MGX_BFS_DIRECTION_OPT, alpha=_alpha(a, n))
        assert torch.equal(torch.from_numpy(bfs.labels()).cuda(), ref), a
        assert st["reached"] == st0["reached"] and st["m_t"] == st0["m_t"]
        pulled += st["pull_edges"]
    assert pulled > 0


def test_config2_rmat22_bfs_full_size_vs_oracle(gpu_ctx, oracle, torch_mod):
    """config 2 at the benchmarked size: RMAT-22 ef 16 symmetrised (n = 4 194 304, m = 134 217 728), hub-first layout and
    unit blocks built by the library, one seeded source: labels bit-exact against the oracle's bfs_problem_t::cpu"""
    import mini_amd
    from mini_amd import rmat
    g = rmat.rmat_csr(gpu_ctx, 22, 16, seed=22)
    assert g["n"] == 4194304 and g["m"] == 134217728
    graph = mini_amd.Graph.from_device(gpu_ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"]).build_layout()
    ro, ci = g["row_offsets"].cpu().numpy(), g["col_indices"].cpu().numpy()
    src = rmat.pick_sources(ro, 3, 22)[2]
    want = oracle.bfs_cpu(ro, ci, src)
    bfs = mini_amd.BfsProblem(graph, src)
    st = bfs.run(src)
    assert np.array_equal(bfs.labels(), want)
    deg = np.diff(ro)
    assert st["reached"] == int((want >= 0).sum()) and st["m_t"] == int(deg[want >= 0].sum())
    assert st["dense_slots"] >= 1          # the big level read its long rows from the unit blocks
    assert st["vshort_slots"] >= 1         # ... and walked its short rows vertex by vertex
    assert st["cold_slots"] >= 1           # ... and its long rows' entries behind the LDS prefix went through the pair lists
    assert st["lazy_slots"] >= 1           # the build behind a heavy push wrote no queues
    st = bfs.run(src, mode=mini_amd.MGX_BFS_DIRECTION_OPT, alpha=4.0)
    assert np.array_equal(bfs.labels(), want)


@pytest.mark.parametrize("scale,undir,env", [
    (20, True, {}), (21, True, {"MGX_BFS_DENSE": "1000000", "MGX_BFS_LAZY": "1048576"}), (20, False, {"MGX_BFS_DENSE": "1000000"}),
    (21, True, {"MGX_BFS_DEFER": "0"}), (20, True, {"MGX_BFS_COLD": "0"}), (21, True, {"MGX_BFS_DENSE": "1000000", "MGX_BFS_MERGED_PUSH": "0"}),
    (21, True, {"MGX_BFS_DENSE": "1000000", "MGX_BFS_HOT_UNITS": "0"}), (21, True, {"MGX_BFS_DENSE": "1000000", "MGX_BFS_COLD": "0"}),
    (21, False, {"MGX_BFS_DENSE": "1000000", "MGX_BFS_PACK24": "0"}),
    (21, True, {"MGX_BFS_COLD": "1", "MGX_BFS_COLD_LISTS": "2", "MGX_BFS_DENSE": "1000000"}), (20, True, {"MGX_BFS_VSHORT": "1000000", "MGX_BFS_DENSE": "0"}),
    (21, False, {"MGX_BFS_VSHORT": "1000000", "MGX_BFS_DENSE": "1000000", "MGX_BFS_LAZY": "1048576", "MGX_BFS_COLD": "1", "MGX_BFS_COLD_LISTS": "2"}),
    (21, True, {"MGX_BFS_VSHORT": "1000000", "MGX_BFS_DENSE": "1000000", "MGX_BFS_CHAIN_MAX_EDGES": "0", "MGX_BFS_COLD": "1", "MGX_BFS_COLD_LISTS": "2"})])
def test_cold_edge_pass_vs_oracle(gpu_ctx, oracle, torch_mod, monkeypatch, scale, undir, env):
    """the cold-edge pass (bfs_fused_cold.hpp) needs a graph with vertices behind the LDS prefix (652 288): R-MAT 20 / 21,
    symmetrised and directed (destinations without out-edges sit at the very end of the hub-first order), default
    thresholds and the unit blocks forced onto every level, bitmaps and replayed marks, the parts launched separately;
    labels against the oracle for hub, ordinary and isolated sources"""
    import mini_amd
    from tests.conftest import skip_unless_lab
    skip_unless_lab(env)                    # (no lab-only switch in the list any more: the short rows' cold lists are in the product since round 5)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n = 1 << scale
    s, d, w = oracle.rmat_edges(scale, 0, 8 * n, 900 + scale, True)
    ro, ci, _ = oracle.csr_from_tuples(n, s, d, None, undir=undir)
    graph = mini_amd.Graph.from_host(gpu_ctx, ro, ci, None).build_layout()
    deg = np.diff(ro)
    # the unit blocks without the lists' entries (what the unit-block body reads when the cold-edge pass runs): built with the lists,
    # fewer units than the full blocks; a run without the pass (MGX_BFS_COLD=0) reads the full blocks and marks those entries itself
    info = graph.layout_info()
    if info["cold_pairs"] > 0 and env.get("MGX_BFS_HOT_UNITS") != "0" and env.get("MGX_BFS_PACK24") != "0":
        assert 0 < info["hot_units"] <= info["units"], info
        assert (info["units"] - info["hot_units"]) * 64 <= info["cold_pairs"], info      # (a unit goes only when 64 entries went)
    else:
        assert info["hot_units"] == 0, info
    rng = np.random.default_rng(scale)
    srcs = [int(np.argmax(deg))] + [int(v) for v in rng.choice(np.where(deg > 0)[0], size=3, replace=False)] + [int(np.where(deg == 0)[0][0])]
    bfs = mini_amd.BfsProblem(graph, srcs[0])
    cold = 0
    for src in srcs:
        want = oracle.bfs_cpu(ro, ci, src)
        st = bfs.run(src)
        assert np.array_equal(bfs.labels(), want), (scale, undir, env, src)
        assert st["m_t"] == int(deg[want >= 0].sum())
        cold += st["cold_slots"]
    if env.get("MGX_BFS_COLD") == "0":
        assert cold == 0
    elif undir and scale >= 21 and env.get("MGX_BFS_DENSE") == "1000000":     # (R-MAT-20: every vertex with edges is inside the prefix)
        assert cold > 0


@pytest.mark.parametrize("env", [{}, {"MGX_BFS_DENSE": "1000000", "MGX_BFS_LAZY": "1048576"}, {"MGX_BFS_DENSE": "1000000", "MGX_BFS_DEFER": "1"},
                                 {"MGX_BFS_DENSE": "1000000", "MGX_BFS_HOT_UNITS": "0"}])
def test_cold_edge_pass_ragged_last_slice(gpu_ctx, oracle, monkeypatch, env):
    """a vertex count that is no multiple of anything (800 003): the last cold slice ends in the middle of a bitmap word and
    of a 1024-vertex run of the queue build; hubs whose neighbours are spread over the whole id range, so that about a
    fifth of the long rows' entries lie behind the LDS prefix"""
    import mini_amd
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(800003)
    n, h = 800003, 1500
    deg = rng.integers(500, 4000, size=h)
    t0 = np.repeat(np.arange(h), deg).astype(np.int32)
    t1 = rng.integers(h, n, size=int(deg.sum())).astype(np.int32)
    e2 = 600000                                                           # ... and a sparse random background
    t0 = np.concatenate([t0, rng.integers(0, n, size=e2).astype(np.int32)])
    t1 = np.concatenate([t1, rng.integers(0, n, size=e2).astype(np.int32)])
    ro, ci, _ = oracle.csr_from_tuples(n, t0, t1, None, undir=True)
    graph = mini_amd.Graph.from_host(gpu_ctx, ro, ci, None).build_layout()
    d = np.diff(ro)
    info = graph.layout_info()
    assert info["cold_pairs"] > 0 and ((0 < info["hot_units"] < info["units"]) if env.get("MGX_BFS_HOT_UNITS") != "0" else info["hot_units"] == 0), info
    bfs = mini_amd.BfsProblem(graph, 0)
    cold = 0
    for src in [0, int(h + 5), int(n - 1), int(np.argmax(d))]:
        want = oracle.bfs_cpu(ro, ci, src)
        st = bfs.run(src)
        assert np.array_equal(bfs.labels(), want), (env, src)
        assert st["m_t"] == int(d[want >= 0].sum())
        cold += st["cold_slots"]
    if env:
        assert cold > 0


def test_rmat23_unit_blocks_and_cold_pass_on_a_big_graph(gpu_ctx, oracle, torch_mod):
    """one size up (n = 8 388 608, m = 268 435 456): the traversal keeps the unit blocks, the cold-edge pass (seven slices
    behind the prefix) and the lazy builds on graphs of this size too -- the bitmap probe of cold neighbours is for
    graphs without unit blocks; labels against the oracle"""
    import mini_amd
    from mini_amd import rmat
    g = rmat.rmat_csr(gpu_ctx, 23, 16, seed=23)
    graph = mini_amd.Graph.from_device(gpu_ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"]).build_layout()
    ro, ci = g["row_offsets"].cpu().numpy(), g["col_indices"].cpu().numpy()
    src = rmat.pick_sources(ro, 2, 23)[1]
    want = oracle.bfs_cpu(ro, ci, src)
    bfs = mini_amd.BfsProblem(graph, src)
    st = bfs.run(src)
    assert np.array_equal(bfs.labels(), want)
    deg = np.diff(ro)
    assert st["m_t"] == int(deg[want >= 0].sum())
    assert st["dense_slots"] >= 1 and st["cold_slots"] >= 1 and st["lazy_slots"] >= 1


@pytest.mark.parametrize("flat_lists", ["1", "0"])
def test_flat_graph_sweeps_all_entries_lists(gpu_ctx, oracle, torch_mod, monkeypatch, flat_lists):
    """round 6: a FLAT graph (uniform random, every row ~32 entries: six endpoints in seven lie behind the 652 288-vertex LDS prefix).
    Its layout carries EVERY entry of every row as a packed (owner, destination) pair by slice of the destination, slices from vertex 0
    on; a level that holds an eighth of the entries is one sweep of those lists by the cold-edge pass's workgroups (nothing else of the
    push grid works), the other levels walk their queues.  Labels and counters against the oracle, against the same traversals with
    the lists switched off (MGX_BFS_FLAT_LISTS=0: the bitmap probes of round 5), one call per source and as a batch."""
    import mini_amd
    from mini_amd import rmat
    monkeypatch.setenv("MGX_BFS_FLAT_LISTS", flat_lists)
    scale = 21
    g = rmat.uniform_csr(gpu_ctx, scale, 16, seed=scale)
    graph = mini_amd.Graph.from_device(gpu_ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"]).build_layout()
    info = graph.layout_info()
    if flat_lists == "1":
        assert info["cold_majority"] == 0 and info["cold_pairs"] == g["m"] and info["cold_slices"] == 4, info      # 2^21 vertices: four slices of 652 288
    else:
        assert info["cold_majority"] == 1 and info["cold_pairs"] == 0, info
    ro, ci = g["row_offsets"].cpu().numpy(), g["col_indices"].cpu().numpy()
    deg = np.diff(ro)
    srcs = [int(s) for s in rmat.pick_sources(ro, 3, scale)]
    bfs = mini_amd.BfsProblem(graph, srcs[0])
    swept = 0
    for s_ in srcs:
        want = oracle.bfs_cpu(ro, ci, s_)
        st = bfs.run(s_)
        assert np.array_equal(bfs.labels(), want), s_
        assert st["m_t"] == int(deg[want >= 0].sum()) and st["levels"] == int(want.max()) + 1, (s_, st)
        swept += st["cold_slots"]
    sts, reruns = bfs.run_many(srcs, mini_amd.MGX_BFS_PUSH, 0.0)
    assert np.array_equal(bfs.labels(), oracle.bfs_cpu(ro, ci, srcs[-1]))
    assert all(x["m_t"] == sts[0]["m_t"] for x in sts)             # one component holds every vertex with an edge
    assert (swept > 0) == (flat_lists == "1"), swept


@pytest.mark.parametrize("hot", ["1", "0"])
def test_hot_unit_blocks_beyond_2_23_vertices(gpu_ctx, oracle, torch_mod, monkeypatch, hot):
    """n = 2^24: the full unit blocks have no 24-bit copy at this size (their entries need 25 bits), the blocks WITHOUT the
    cold-edge lists' entries do (what is left points into the LDS prefix) -- the unit-block body reads those, 3 bytes per entry,
    whenever the cold-edge pass runs; MGX_BFS_HOT_UNITS=0: the full blocks at 4 bytes.  Labels against the oracle either way,
    push and direction-optimising."""
    import mini_amd
    from mini_amd import rmat
    monkeypatch.setenv("MGX_BFS_HOT_UNITS", hot)
    monkeypatch.setenv("MGX_BFS_DENSE", "1000000")          # (the unit blocks on every level whose frontier is a bitmap)
    g = rmat.rmat_csr(gpu_ctx, 24, 3, seed=124)
    graph = mini_amd.Graph.from_device(gpu_ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"]).build_layout()
    info = graph.layout_info()
    assert info["units"] > 0 and info["units_24bit"] == 0 and info["cold_pairs"] > 0, info
    assert (info["hot_units"] > 0) == (hot == "1"), info
    ro, ci = g["row_offsets"].cpu().numpy(), g["col_indices"].cpu().numpy()
    deg = np.diff(ro)
    bfs = mini_amd.BfsProblem(graph, 0)
    cold = 0
    for src in rmat.pick_sources(ro, 2, 124) + [int(np.argmax(deg))]:
        want = oracle.bfs_cpu(ro, ci, src)
        st = bfs.run(src)
        assert np.array_equal(bfs.labels(), want), (hot, src)
        assert st["m_t"] == int(deg[want >= 0].sum())
        cold += st["cold_slots"]
        bfs.run(src, mode=mini_amd.MGX_BFS_DIRECTION_OPT, alpha=4.0)
        assert np.array_equal(bfs.labels(), want), (hot, src, "direction-optimising")
    assert cold > 0


def test_config3_rmat22_sssp_full_size_vs_oracle(gpu_ctx, oracle, torch_mod):
    """config 3 at the benchmarked size: the same topology with integer weights in [0, 63] (every float32 path sum is
    exact, so the north star's 1e-6 relative tolerance is met with equality): fused SSSP distances against the oracle's
    independent float Dijkstra"""
    import mini_amd
    from mini_amd import rmat
    g = rmat.rmat_csr(gpu_ctx, 22, 16, seed=22, weighted=True)
    graph = mini_amd.Graph.from_device(gpu_ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"], g["weights"])
    graph.build_layout(weights=True)
    ro, ci, w = g["row_offsets"].cpu().numpy(), g["col_indices"].cpu().numpy(), g["weights"].cpu().numpy()
    src = rmat.pick_sources(ro, 1, 22)[0]
    want = oracle.sssp_dijkstra_f32(ro, ci, w, src)
    sssp = mini_amd.SsspProblem(graph, src)
    sssp.run(src)
    got = sssp.distances()
    assert np.array_equal(got, want)            # (unreachable: FLT_MAX on both sides)
    assert int((want < np.finfo(np.float32).max).sum()) > g["n"] // 3


def test_real_dataset_if_supplied(gpu_ctx, oracle, torch_mod):
    """soc-LiveJournal1 (config 4's named input) is not available offline; when MGX_DATA_DIR holds
    soc-LiveJournal1.mtx (or MGX_DATASET names any MTX file) it goes through the product's own loader, a library-built
    CSC, the fused push and direction-optimising runs and the oracle (one source)."""
    import mini_amd
    path = os.environ.get("MGX_DATASET") or os.path.join(os.environ.get("MGX_DATA_DIR", "/nonexistent"), "soc-LiveJournal1.mtx")
    if not os.path.exists(path):
        pytest.skip("no dataset supplied (MGX_DATA_DIR / MGX_DATASET)")
    n, ro, ci, w = mini_amd.load_mtx(path, undir=False)
    g = mini_amd.Graph.from_host(gpu_ctx, ro, ci, w).build_csc()
    deg = np.diff(ro)
    src = int(np.argmax(deg))
    want = oracle.bfs_cpu(ro, ci, src)
    bfs = mini_amd.BfsProblem(g, src)
    bfs.run(src)
    assert np.array_equal(bfs.labels(), want)
    for a in ALPHAS:
        bfs.run(src, mode=mini_amd.MGX_BFS_DIRECTION_OPT, alpha=_alpha(a, n))
        assert np.array_equal(bfs.labels(), want), a


# ---- config 5: the partitioned engine at its named size, G rank engines in turn on one GPU ---------------------------
def _rank_engines(ctx, torch, scale, ranks, seed):
    from mini_amd.dist_bfs import HipRankEngine2, rmat_cyclic_shard
    dev = torch.device("cuda", 0)
    engs, shards = [], []
    for r in range(ranks):
        ro, col, new_of_old, old_of_new, deg_new = rmat_cyclic_shard(ctx, scale, 16, seed, ranks, r, dev)
        engs.append(HipRankEngine2(ctx, 1 << scale, ranks, r, ro, col))
        shards.append((ro, col))
    return engs, shards, new_of_old, old_of_new, deg_new


def _run_rank_engines(torch, engs, src, hint=8):
    """what tools/dist2_single.py does: every level's push on every rank, the all-gather of the new-bit maps as a
    concatenation, every rank merges all maps; the host looks after `hint` levels, then every two"""
    G = len(engs)
    for e in engs:
        e.reset(src)
    level, batch = 0, hint
    while True:
        for _ in range(batch):
            gathered = torch.cat([e.push(level) for e in engs]) if G > 1 else engs[0].push(level)
            for e in engs:
                e.merge(level, gathered, G)
            level += 1
        sts = [e.status(level) for e in engs]
        if sts[0]["over"]:
            return sts
        batch = 2


def _gathered_labels(engs, n):
    G = len(engs)
    lab = np.empty(n, dtype=np.int32)
    for r, e in enumerate(engs):
        lab[r::G] = e.labels()                        # vertex v = local row * G + rank
    return lab


def _check_partitioned_traversal(torch, engs, shards, deg_new, src, sts):
    from bench_dist import _tree_check_local
    G, n = len(engs), engs[0].n_global
    assert all(st["over"] for st in sts) and len({st["levels"] for st in sts}) == 1
    lab = _gathered_labels(engs, n)
    reached = lab >= 0
    assert lab[src] == 0 and int(lab.max()) + 1 == sts[0]["levels"]
    # the edges the ranks expanded: every entry of every reached vertex's row, once
    assert sum(st["edges_local"] for st in sts) == int(deg_new.cpu().numpy()[reached].sum())
    # every rank ends with the same visited bitmap, and it is the reached set
    want_bits = np.packbits(reached, bitorder="little")
    want_words = np.zeros(engs[0].nwords, dtype=np.uint32)
    want_words.view(np.uint8)[: len(want_bits)] = want_bits
    for e in engs:
        assert np.array_equal(e.visited(), want_words), "rank %d's bitmap differs from the reached set" % e.rank
    # BFS-tree properties over every rank's own rows (the graph is symmetric: a row lists all neighbours)
    lab_dev = torch.from_numpy(lab).cuda()
    for r, (ro, col) in enumerate(shards):
        assert _tree_check_local(lab_dev, ro, col, G, r, src), "rank %d: BFS-tree properties violated" % r
    return lab


def _run_rank_engines_lists(torch, engs, src):
    """the level protocol of mgx_dbfs2_run / DistBfs2.run: id lists first (the all-gather a concatenation, every engine's list
    merge with its verdict), the bitmaps only when some rank's list overflowed; -> (statuses, levels merged from lists, from bitmaps)"""
    G = len(engs)
    for e in engs:
        e.reset(src)
    level = sparse = dense = 0
    while True:
        maps = [e.push(level) for e in engs]
        glists = torch.cat([e.list for e in engs]) if G > 1 else engs[0].list
        res = [e.apply_lists(level, glists, G) for e in engs]
        assert len(set(res)) == 1, "the ranks disagree about the level's lists"
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
    return [e.status(level) for e in engs], sparse, dense


# round 4's rank engine: every new path against its switch, in ONE process (the engines read the switches when they
# are made), on graphs whose id range outgrows the LDS prefix (R-MAT 21: cold-edge lists), with the unit-block and
# vertex-by-vertex bodies forced onto every level that can take them
_RANK_SWITCHES = [
    {},                                                                  # the defaults
    {"MGX_DIST_DENSE_DIV": "1000000", "MGX_DIST_VSHORT": "1000000", "MGX_DIST_DEFER": "2"},     # dense bodies wherever the frontier bitmap is current; deferred hot marks whatever the shard's size
    {"MGX_DIST_DENSE_DIV": "1000000", "MGX_DIST_VSHORT": "1000000", "MGX_DIST_DECLARE_MUL": "1", "MGX_DIST_DEFER": "2"},   # lists declared overflowed early
    {"MGX_DIST_DENSE_DIV": "1000000", "MGX_DIST_SPARSE_PUSH": "0", "MGX_DIST_FUSED_MERGE": "0"},
    {"MGX_DIST_DENSE_DIV": "1000000", "MGX_DIST_COLD_REDUCE": "0", "MGX_DIST_HOT_UNITS": "0"},
    {"MGX_DIST_DENSE_DIV": "1000000", "MGX_BFS_COLD_PACK": "0", "MGX_DIST_DEFER": "0", "MGX_DIST_VSHORT": "0"},
    {"MGX_DIST_DENSE_DIV": "1000000", "MGX_DIST_VSHORT": "1000000", "MGX_BFS_PACK24": "0", "MGX_DIST_COLD_WGS": "700"},
]


@pytest.mark.parametrize("scale,G", [(21, 8), (21, 2), (19, 4)])
def test_rank_engine_round4_paths_vs_single_gpu(gpu_ctx, torch_mod, monkeypatch, scale, G):
    """sparse levels appended by the push, declared overflows, the OR-merge inside the queue build (2 / 4 / 8 ranks), the stream
    reduce of the cold and deferred bitmaps, hot-only 24-bit unit blocks, 4-byte cold-edge pairs, deferred hot marks, short rows
    vertex by vertex with the cold test: under both level protocols the ranks' labels equal the single-GPU traversal of the
    unpartitioned graph, vertex by vertex, and every rank ends with the same bitmap"""
    import mini_amd
    from mini_amd import rmat
    torch = torch_mod
    g = rmat.rmat_csr(gpu_ctx, scale, 16, seed=scale)
    graph = mini_amd.Graph.from_device(gpu_ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"]).build_layout()
    single = {}
    bfs = None
    for env in _RANK_SWITCHES:
        with monkeypatch.context() as mp:
            for k, v in env.items():
                mp.setenv(k, v)
            engs, shards, new_of_old, old_of_new, deg_new = _rank_engines(gpu_ctx, torch, scale, G, scale)
        o2n, n2o = old_of_new.cpu().numpy(), new_of_old.cpu().numpy()
        cand = torch.nonzero(deg_new > 0)[:, 0]
        srcs = [int(cand[0]), int(cand[len(cand) // 2]), int(cand[-1])]     # the biggest hub, a middling vertex, a leaf
        for src in srcs:
            if src not in single:
                src_old = int(o2n[src])
                bfs = bfs or mini_amd.BfsProblem(graph, src_old)
                bfs.run(src_old)
                single[src] = bfs.labels().copy()
            for proto in ("lists", "bitmaps"):
                if proto == "lists":
                    sts, sparse, dense = _run_rank_engines_lists(torch, engs, src)
                    assert sparse >= 1, "no level was merged from id lists: %r" % (env,)
                else:
                    sts = _run_rank_engines(torch, engs, src)
                lab_new = _check_partitioned_traversal(torch, engs, shards, deg_new, src, sts)
                assert np.array_equal(lab_new[n2o], single[src]), "labels differ from the single-GPU traversal: %r, %s, source %d" % (env, proto, src)
                # ... and the paths the switches ask for were the ones that ran
                paths = [e.path_levels() for e in engs]
                if env.get("MGX_DIST_SPARSE_PUSH") == "0":
                    assert all(p[0] == 0 for p in paths)
                else:
                    assert all(p[0] >= 1 for p in paths), "no level was appended by the push: %r" % (paths,)
                forced = env.get("MGX_DIST_DENSE_DIV") == "1000000"
                if forced and scale >= 21:
                    assert all(e.dense_levels() >= 1 and e.cold_levels()[0] >= 1 for e in engs), "the cold-edge pass did not run"
                    assert all((p[3] > 0) == (env.get("MGX_BFS_COLD_PACK") != "0") for p in paths), "packed pair lists: %r" % (paths,)
                if env.get("MGX_DIST_VSHORT") == "1000000":
                    assert any(p[1] >= 1 for p in paths), "no level walked its short rows vertex by vertex: %r" % (paths,)
                if env.get("MGX_DIST_VSHORT") == "0":
                    assert all(p[1] == 0 for p in paths)
                if env.get("MGX_DIST_DECLARE_MUL") == "1" and proto == "lists":
                    assert any(p[2] == 1 for p in paths), "no list was declared overflowed: %r" % (paths,)
        for e in engs:
            e.close()


def test_config5_rmat26_eight_rank_engines(gpu_ctx, torch_mod):
    """BASELINE config 5's workload: RMAT-26 ef 16 (n = 67 108 864, 2 147 483 648 entries), cyclic partition over 8 ranks"""
    torch = torch_mod
    scale, G = 26, 8
    engs, shards, new_of_old, old_of_new, deg_new = _rank_engines(gpu_ctx, torch, scale, G, scale)
    assert sum(int(ro[-1]) for ro, _ in shards) == 2 * 16 * (1 << scale)          # 64-bit global count, int32 per rank
    assert all(e.units > 0 for e in engs) and all(e.cold_levels()[1] > 0 for e in engs)   # unit blocks + cold-edge lists per rank
    cand = torch.nonzero(deg_new > 0)[:, 0]
    srcs = [int(cand[0]), int(cand[len(cand) // 3]), int(cand[-1])]       # the biggest hub, a middling vertex, a leaf
    for src in srcs:
        sts = _run_rank_engines(torch, engs, src)
        lab = _check_partitioned_traversal(torch, engs, shards, deg_new, src, sts)
        assert (lab >= 0).sum() > (1 << scale) // 3                       # the giant component
    for e in engs:
        e.close()


def test_config5_shape_rmat23_four_rank_engines_vs_single_gpu_and_oracle(gpu_ctx, oracle, torch_mod):
    """the same engine at a size the single-GPU path and the oracle can check vertex by vertex"""
    import mini_amd
    from mini_amd import rmat
    torch = torch_mod
    scale, G = 23, 4
    engs, shards, new_of_old, old_of_new, deg_new = _rank_engines(gpu_ctx, torch, scale, G, scale)
    g = rmat.rmat_csr(gpu_ctx, scale, 16, seed=scale)
    graph = mini_amd.Graph.from_device(gpu_ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"]).build_layout()
    o2n = old_of_new.cpu().numpy()
    cand = torch.nonzero(deg_new > 0)[:, 0]
    bfs = None
    for i, src in enumerate((int(cand[1]), int(cand[len(cand) // 2]))):
        sts = _run_rank_engines(torch, engs, src)
        lab_new = _check_partitioned_traversal(torch, engs, shards, deg_new, src, sts)
        src_old = int(o2n[src])
        bfs = bfs or mini_amd.BfsProblem(graph, src_old)
        bfs.run(src_old)
        single = bfs.labels()
        assert np.array_equal(lab_new[new_of_old.cpu().numpy()], single)
        if i == 0:
            want = oracle.bfs_cpu(g["row_offsets"].cpu().numpy(), g["col_indices"].cpu().numpy(), src_old)
            assert np.array_equal(single, want)
    for e in engs:
        e.close()


def test_rmat26_on_one_gpu_as_two_shards(gpu_ctx):
    """the 1-GPU figure for BASELINE config 5's graph (the denominator of the north star's ">= 5x at 8 GPUs over 1 GPU on
    RMAT-26"): 2^31 CSR entries do not fit int32 row offsets, so `bench.py --scale 26 --gpus 1` cuts the graph into
    two cyclic shards of 2^30 entries, a rank engine each, and runs them in turn through the partitioned traversal's C++ loop
    (mgx_dbfs2_run_group).  The line is only printed with parity true: BFS-tree properties of the labels over every shard's rows."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--scale", "26", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["scale"] == 26 and line["parity_vs_oracle"] is True
    assert line["config"]["parallelism"] == "1 GPU, 2 shards in turn"
    assert line["value"] > 0 and line["avg_levels"] >= 5
