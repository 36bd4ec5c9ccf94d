"""GPU suite (-m gpu): the HIP path, called through the C-ABI, against the CPU oracle on the same
seeded inputs, the committed golden fixtures, and size-independent properties at large sizes.
Bar: bit-exact for every integer array; SSSP distances bit-exact (weights are small integers, so
every float32 path sum is exact -- tolerance stated by the north star is 1e-6 relative)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = json.load(open(os.path.join(GOLD, "reference_goldens.json")))["cases"]
FLT_MAX = np.finfo(np.float32).max


def _graph(ctx, ro, ci, w=None, csc=None):
    import mini_amd
    if csc is None:
        return mini_amd.Graph.from_host(ctx, ro, ci, w)
    return mini_amd.Graph.from_host(ctx, ro, ci, w, csc[0], csc[1])


@pytest.fixture(scope="module")
def rmat_graphs(oracle):
    """oracle-built CSRs (host) for several scales; seeded"""
    out = {}
    for scale, seed in ((8, 1), (10, 10), (13, 13), (16, 16)):
        out[scale] = oracle.rmat_csr(scale, 16, seed)
    return out


# ---- generator -----------------------------------------------------------------------------
@pytest.mark.parametrize("scale,seed,scramble", [(6, 3, True), (10, 10, True), (10, 10, False), (17, 99, True)])
def test_rmat_generator_matches_spec(gpu_ctx, oracle, torch_mod, scale, seed, scramble):
    import mini_amd
    torch = torch_mod
    count, first = 5000, 12345
    s = torch.empty(count, dtype=torch.int32, device="cuda")
    d = torch.empty(count, dtype=torch.int32, device="cuda")
    w = torch.empty(count, dtype=torch.float32, device="cuda")
    mini_amd.rmat_edges(gpu_ctx, scale, first, count, seed, scramble, s, d, w)
    gpu_ctx.synchronize()
    es, ed, ew = oracle.rmat_edges(scale, first, count, seed, scramble)
    assert np.array_equal(s.cpu().numpy(), es)
    assert np.array_equal(d.cpu().numpy(), ed)
    assert np.array_equal(w.cpu().numpy(), ew)


def test_rmat_csr_builder_matches_oracle_loader_semantics(gpu_ctx, oracle):
    from mini_amd import rmat
    g = rmat.rmat_csr(gpu_ctx, scale=11, edgefactor=16, seed=11, weighted=True)
    n, ro, ci, w = oracle.rmat_csr(11, 16, 11)
    assert g["n"] == n and g["m"] == len(ci)
    assert np.array_equal(g["row_offsets"].cpu().numpy(), ro)
    assert np.array_equal(g["col_indices"].cpu().numpy(), ci)
    assert np.array_equal(g["weights"].cpu().numpy(), w)


# ---- building blocks -----------------------------------------------------------------------
@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 2047, 2048, 2049, 100000, 1 << 21])
def test_scan_exclusive(gpu_ctx, torch_mod, n):
    import mini_amd
    torch = torch_mod
    rng = np.random.default_rng(n)
    h = rng.integers(0, 1000, size=max(n, 1), dtype=np.int32)[:n]
    d_in = torch.from_numpy(np.ascontiguousarray(h)).cuda() if n else torch.empty(1, dtype=torch.int32, device="cuda")
    d_out = torch.empty(max(n, 1), dtype=torch.int32, device="cuda")
    total = mini_amd.scan_exclusive_i32(gpu_ctx, d_in, n, d_out)
    want = np.concatenate([[0], np.cumsum(h, dtype=np.int64)])
    assert total == want[-1]
    assert np.array_equal(d_out.cpu().numpy()[:n], want[:-1].astype(np.int32))


@pytest.mark.parametrize("n,keep", [(0, 0.5), (1, 1.0), (64, 0.0), (2049, 0.3), (1 << 20, 0.02), (1 << 20, 0.97)])
def test_compact_is_stable_and_exact(gpu_ctx, torch_mod, n, keep):
    import mini_amd
    torch = torch_mod
    rng = np.random.default_rng(n + int(keep * 100))
    h = rng.integers(0, 1 << 30, size=max(n, 1), dtype=np.int32)[:n]
    h[rng.random(n) >= keep] = -1
    d_in = torch.from_numpy(np.ascontiguousarray(h)).cuda() if n else torch.empty(1, dtype=torch.int32, device="cuda")
    d_out = torch.full((max(n, 1),), -7, dtype=torch.int32, device="cuda")
    kept = mini_amd.compact_i32(gpu_ctx, d_in, n, -1, d_out)
    want = h[h != -1]
    assert kept == len(want)
    assert np.array_equal(d_out.cpu().numpy()[:kept], want)


def test_lbs_enumeration_with_empty_and_huge_segments(gpu_ctx, oracle):
    import mini_amd
    # degrees: a hub of 5000, runs of empty rows, small rows
    deg = np.array([0, 0, 5000, 1, 0, 0, 0, 3, 2000, 0, 1, 1, 1, 0, 700], dtype=np.int64)
    ro = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    ci = np.zeros(ro[-1], dtype=np.int32)
    g = _graph(gpu_ctx, ro, ci)
    rng = np.random.default_rng(5)
    ids = np.concatenate([rng.integers(0, len(deg), size=300), np.zeros(1500, dtype=np.int64),
                          [2, 8, 2]]).astype(np.int32)
    f = mini_amd.Frontier(gpu_ctx, len(ids)).load(ids)
    total = mini_amd.scan_frontier_degrees(g, f)
    scanned, want_total = oracle.scan_degrees(ro, ids)
    assert total == want_total
    seg, rank = mini_amd.lbs_expand_debug(g, f, total)
    wseg, wrank = oracle.lbs(scanned, want_total)
    assert np.array_equal(seg, wseg) and np.array_equal(rank, wrank)
    # empty frontier / frontier of isolated vertices
    f0 = mini_amd.Frontier(gpu_ctx, 4).load(np.array([0, 1, 4], dtype=np.int32))
    assert mini_amd.scan_frontier_degrees(g, f0) == 0
    f1 = mini_amd.Frontier(gpu_ctx, 4).load(np.zeros(0, dtype=np.int32))
    assert mini_amd.scan_frontier_degrees(g, f1) == 0


@pytest.mark.parametrize("op", ["f32_plus", "i32_min", "i32_max"])
def test_neighbour_reduce_matches_oracle(gpu_ctx, oracle, torch_mod, rmat_graphs, op):
    import mini_amd
    torch = torch_mod
    n, ro, ci, w = rmat_graphs[13]
    g = _graph(gpu_ctx, ro, ci)
    rng = np.random.default_rng(3)
    ids = rng.permutation(n)[: n // 2].astype(np.int32)
    ids[:3] = int(np.argmax(np.diff(ro)))          # the hub three times: segments that span tiles
    f = mini_amd.Frontier(gpu_ctx, n).load(ids)
    if op == "f32_plus":
        vals = rng.integers(0, 8, size=n).astype(np.float32)   # small ints: float sums are exact
        dv = torch.from_numpy(vals).cuda()
        red = torch.full((len(ids),), -1, dtype=torch.float32, device="cuda")
        nz = mini_amd.segreduce(g, f, dv, 0.0, red, op)
        want, wnz = oracle.neighbor_reduce_f32_plus(ro, ci, ids, vals, 0.0)
    else:
        vals = rng.integers(-1000, 1000, size=n).astype(np.int32)
        dv = torch.from_numpy(vals).cuda()
        ident = 2**31 - 1 if op == "i32_min" else -2**31
        red = torch.full((len(ids),), 12345, dtype=torch.int32, device="cuda")
        nz = mini_amd.segreduce(g, f, dv, ident, red, op)
        want, wnz = oracle.neighbor_reduce_i32(ro, ci, ids, vals, ident, op == "i32_max")
    assert nz == wnz
    assert np.array_equal(red.cpu().numpy(), want)


@pytest.mark.parametrize("op", ["f32_plus", "i32_min", "i32_max"])
@pytest.mark.parametrize("scale,ef", [(9, 4), (13, 16), (16, 16)])
def test_neighbour_reduce_full_frontier_on_the_layout(gpu_ctx, oracle, torch_mod, op, scale, ef):
    """the full-frontier path of the neighbour-reduce (mgx/nreduce.hpp: unit blocks for the long rows, degree classes for the
    short ones, hub values in LDS) -- taken when the frontier is 0 .. n - 1 and the graph carries the library's hub-first
    layout -- against the oracle's serial restatement: exact for int min / max and for float sums of small integers, 2e-5
    relative for real-valued floats (the fold order differs); a permuted frontier of the same size takes the general
    kernel and is compared the same way; R-MAT 16 has rows of more than 4096 entries (a workgroup per row)"""
    import mini_amd
    torch = torch_mod
    n, ro, ci, w = oracle.rmat_csr(scale, ef, 60 + scale)
    g = _graph(gpu_ctx, ro, ci).build_layout()
    rng = np.random.default_rng(scale)
    for frontier_kind in ("iota", "permuted"):
        ids = np.arange(n, dtype=np.int32) if frontier_kind == "iota" else rng.permutation(n).astype(np.int32)
        f = mini_amd.Frontier(gpu_ctx, n).load(ids)
        if op == "f32_plus":
            for real in (False, True):
                vals = (rng.random(n) * 3.0).astype(np.float32) if real else rng.integers(0, 8, size=n).astype(np.float32)
                dv = torch.from_numpy(vals).cuda()
                red = torch.full((n,), -1, dtype=torch.float32, device="cuda")
                nz = mini_amd.segreduce(g, f, dv, 0.0, red, op)
                want, wnz = oracle.neighbor_reduce_f32_plus(ro, ci, ids, vals, 0.0)
                assert nz == wnz == len(ci)
                got = red.cpu().numpy()
                if real:
                    assert np.allclose(got, want, rtol=2e-5, atol=1e-6), (frontier_kind, np.abs(got - want).max())
                else:
                    assert np.array_equal(got, want), frontier_kind
        else:
            vals = rng.integers(-1000, 1000, size=n).astype(np.int32)
            dv = torch.from_numpy(vals).cuda()
            ident = 2**31 - 1 if op == "i32_min" else -2**31
            red = torch.full((n,), 12345, dtype=torch.int32, device="cuda")
            nz = mini_amd.segreduce(g, f, dv, ident, red, op)
            want, wnz = oracle.neighbor_reduce_i32(ro, ci, ids, vals, ident, op == "i32_max")
            assert nz == wnz
            assert np.array_equal(red.cpu().numpy(), want), frontier_kind


@pytest.mark.parametrize("op", ["f32_plus", "i32_min", "i32_max"])
@pytest.mark.parametrize("scale,ef,slices", [(9, 4, 0), (13, 16, 0), (16, 16, 0), (17, 16, 2)])
def test_neighbour_reduce_subset_frontier_on_the_layout(gpu_ctx, oracle, torch_mod, monkeypatch, op, scale, ef, slices):
    """round 6: a frontier that is a large ASCENDING SUBSET of the vertices (what PR's filter leaves behind its first iteration,
    pr_enactor.hxx:53-66) takes the layout's kernels too -- every row is computed, the frontier's are kept, reduced[] is indexed by
    frontier POSITION (neighborhood.hxx:58) -- against the oracle's serial restatement, like the full frontier.  Also: a subset
    below n / 8 (general kernel), ids out of order and an ascending list with a duplicate (both: the device-side check sends
    them to the general kernel, same answers), a frontier that holds every vertex but one, vertices without edges inside the frontier (identity), and MGX_NR_SLICES=2 on R-MAT 17 so that the
    tail behind the hot slices is used."""
    import mini_amd
    torch = torch_mod
    if slices:
        monkeypatch.setenv("MGX_NR_SLICES", str(slices))
    n, ro, ci, w = oracle.rmat_csr(scale, ef, 160 + scale)
    g = _graph(gpu_ctx, ro, ci).build_layout()
    rng = np.random.default_rng(100 + scale)
    deg = np.diff(ro)
    kinds = {
        "half": np.sort(rng.permutation(n)[: n // 2]),
        "with_edges": np.nonzero(deg > 0)[0],                       # PR's second frontier, nearly
        "all_but_one": np.delete(np.arange(n), n // 3),
        "eighth": np.sort(rng.permutation(n)[: (n + 7) // 8]),      # exactly at the threshold
        "sparse": np.sort(rng.permutation(n)[: n // 50]),           # below it: the general kernel
        "shuffled": rng.permutation(n)[: n // 2],                   # not ascending: the general kernel
    }
    dup = np.sort(rng.permutation(n)[: n // 2]); dup[5] = dup[4]    # a duplicate: not STRICTLY ascending
    kinds["duplicate"] = dup
    for kind, ids in kinds.items():
        ids = np.ascontiguousarray(ids, dtype=np.int32)
        f = mini_amd.Frontier(gpu_ctx, n).load(ids)
        if op == "f32_plus":
            for real in (False, True):
                vals = (rng.random(n) * 3.0).astype(np.float32) if real else rng.integers(0, 8, size=n).astype(np.float32)
                dv = torch.from_numpy(vals).cuda()
                red = torch.full((len(ids),), -1, dtype=torch.float32, device="cuda")
                nz = mini_amd.segreduce(g, f, dv, 0.0, red, op)
                want, wnz = oracle.neighbor_reduce_f32_plus(ro, ci, ids, vals, 0.0)
                assert nz == wnz, kind
                got = red.cpu().numpy()
                if real:
                    assert np.allclose(got, want, rtol=2e-5, atol=1e-6), (kind, np.abs(got - want).max())
                else:
                    assert np.array_equal(got, want), kind
        else:
            vals = rng.integers(-1000, 1000, size=n).astype(np.int32)
            dv = torch.from_numpy(vals).cuda()
            ident = 2**31 - 1 if op == "i32_min" else -2**31
            red = torch.full((len(ids),), 12345, dtype=torch.int32, device="cuda")
            nz = mini_amd.segreduce(g, f, dv, ident, red, op)
            want, wnz = oracle.neighbor_reduce_i32(ro, ci, ids, vals, ident, op == "i32_max")
            assert nz == wnz, kind
            assert np.array_equal(red.cpu().numpy(), want), kind
    # two calls in a row on the same context: the second frontier's positions must not be mixed up with the first one's
    a_ids, b_ids = np.ascontiguousarray(kinds["half"], dtype=np.int32), np.ascontiguousarray(kinds["with_edges"], dtype=np.int32)
    vals = rng.integers(0, 8, size=n).astype(np.int32)
    dv = torch.from_numpy(vals).cuda()
    for ids in (a_ids, b_ids, a_ids):
        f = mini_amd.Frontier(gpu_ctx, n).load(ids)
        red = torch.full((len(ids),), 777, dtype=torch.int32, device="cuda")
        mini_amd.segreduce(g, f, dv, 2**31 - 1, red, "i32_min")
        want, _ = oracle.neighbor_reduce_i32(ro, ci, ids, vals, 2**31 - 1, False)
        assert np.array_equal(red.cpu().numpy(), want)


# ---- golden fixtures through the C-ABI -------------------------------------------------------
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_reference_fixtures_bfs_and_sssp(gpu_ctx, oracle, case, tmp_path):
    """the HIP path DIRECTLY against the reference's own outputs (tests/golden/reference_goldens.json, regenerated from
    /root/reference by tools/regen_goldens.sh): its fixtures from src 0, simple R-MAT graphs of 1 K .. 16 K vertices from
    their busiest row -- labels bit-exact, distances equal to the reference validator's int distances"""
    import mini_amd
    from tests.golden_inputs import case_path, matches
    n, ro, ci, w = mini_amd.load_mtx(case_path(case, oracle, tmp_path, GOLD), undir=case["undir"])
    src = case["src"]
    g = _graph(gpu_ctx, ro, ci, w)
    want = oracle.bfs_cpu(ro, ci, src)
    assert matches(case, "bfs_labels", want, np.int32)                # (the oracle itself against the golden, once more)
    bfs = mini_amd.BfsProblem(g, src)
    st = bfs.run(src)
    assert matches(case, "bfs_labels", bfs.labels(), np.int32)
    assert st["reached"] == int((want >= 0).sum())
    assert st["m_t"] == int(np.diff(ro)[want >= 0].sum())
    assert np.all(bfs.preds() == -1)
    if len(ci) >= 1:
        bfs.reset(src)
        bfs.enact_pushpull()                                          # alpha = 1/n (test_bfs.cu:30)
        assert matches(case, "bfs_labels", bfs.labels(), np.int32)
        if case["undir"] and len(ci) >= n:
            for alpha in (0.5, 4.0):
                bfs.reset(src)
                bfs.enact_pushpull(alpha)
                assert matches(case, "bfs_labels", bfs.labels(), np.int32), alpha
            if "gen" in case:                                         # the bench path on the library's hub-first layout
                g.build_layout(weights=True)
                bfs.run(src)
                assert matches(case, "bfs_labels", bfs.labels(), np.int32)
                bfs.run(src, mini_amd.MGX_BFS_DIRECTION_OPT, 4.0)
                assert matches(case, "bfs_labels", bfs.labels(), np.int32)
    # SSSP distances == the reference CPU validator's int distances (exact), unreachable = FLT_MAX
    if len(ci):
        sssp = mini_amd.SsspProblem(g, src)
        sssp.enact(1.5)
        dist = sssp.distances()
        unreached = dist == FLT_MAX
        as_int = np.where(unreached, 0.0, dist).astype(np.int32)
        as_int[unreached] = np.iinfo(np.int32).max
        assert matches(case, "sssp_dist", as_int, np.int32)
        sssp.run(src)                                                 # the fused device-resident loop: same fixed point
        assert np.array_equal(sssp.distances(), dist)


# ---- operators, one superstep at a time --------------------------------------------------------
def test_bfs_operators_superstep_by_superstep(gpu_ctx, oracle, rmat_graphs):
    import mini_amd
    n, ro, ci, w = rmat_graphs[10]
    m = len(ci)
    g = _graph(gpu_ctx, ro, ci)
    src = int(np.argmax(np.diff(ro)))
    bfs = mini_amd.BfsProblem(g, src)
    fa, fb = mini_amd.Frontier(gpu_ctx, m), mini_amd.Frontier(gpu_ctx, m)
    fa.load(np.array([src], dtype=np.int32))
    o_labels = np.full(n, -1, dtype=np.int32)
    o_labels[src] = 0
    o_front = np.array([src], dtype=np.int32)
    for it in range(32):
        front = bfs.advance(fa, fb, it)
        raw = fb.read()
        o_raw = oracle.bfs_advance(ro, ci, o_labels, o_front, it)
        assert front == len(o_raw) == len(raw)
        # which duplicate edge wins the CAS is schedule dependent; the SET of winners and the
        # positions of -1 for already-labelled targets are not
        assert set(raw[raw >= 0].tolist()) == set(o_raw[o_raw >= 0].tolist())
        assert len(raw[raw >= 0]) == len(o_raw[o_raw >= 0])            # exactly one winner each
        assert np.array_equal(bfs.labels(), o_labels)
        if front == 0:
            break
        kept = bfs.filter(fb, fa, it)
        out = fa.read()
        assert kept == len(out)
        assert np.array_equal(out, raw[raw != -1])                      # stable compaction
        o_front = np.sort(oracle.bfs_filter(o_raw))
        assert np.array_equal(np.sort(out), o_front)
        if kept == 0:
            break
        fa.load(o_front)     # same order on both sides from here
    assert np.array_equal(bfs.labels(), oracle.bfs_cpu(ro, ci, src))


@pytest.mark.parametrize("disturb", ["grow_arena", "overwrite", "expose"])
def test_filter_does_not_trust_stale_keep_ballots(gpu_ctx, oracle, rmat_graphs, disturb):
    """the advance leaves its keep-ballots in the context's scratch arena for the filter behind it.  Whatever
    happens in between -- the arena reallocated (a bigger frontier is created), the frontier overwritten through the C-ABI,
    its device pointer handed out -- the filter must then evaluate cond_filter itself: the result equals the predicate on
    what the frontier holds NOW."""
    import mini_amd
    n, ro, ci, w = rmat_graphs[13]
    m = len(ci)
    g = _graph(gpu_ctx, ro, ci)
    src = int(np.argmax(np.diff(ro)))
    bfs = mini_amd.BfsProblem(g, src)
    fa, fb = mini_amd.Frontier(gpu_ctx, m), mini_amd.Frontier(gpu_ctx, m)
    fa.load(np.array([src], dtype=np.int32))
    front = bfs.advance(fa, fb, 0)
    raw = fb.read()
    assert front == len(raw) and front > 64
    keepalive = None
    if disturb == "grow_arena":
        keepalive = mini_amd.Frontier(gpu_ctx, 40 * m)          # scan scratch for 40 m items: the arena is reallocated
        now = raw
    elif disturb == "overwrite":
        now = raw.copy()
        now[::2] = -1                                           # same pointer, same size, same iteration: other contents
        now[1::2] = np.arange(len(now[1::2]), dtype=np.int32) % n
        fb.load(now)
    else:
        assert fb.device_ptr                                     # from here on anybody may write into it
        now = raw
    kept = bfs.filter(fb, fa, 0)
    out = fa.read()
    assert kept == len(out)
    assert np.array_equal(out, now[now != -1])
    del keepalive


def test_bfs_fused_operator_equals_advance_plus_filter(gpu_ctx, oracle, rmat_graphs):
    import mini_amd
    n, ro, ci, w = rmat_graphs[13]
    g = _graph(gpu_ctx, ro, ci)
    src = int(np.argmax(np.diff(ro)))
    bfs = mini_amd.BfsProblem(g, src)
    fa, fb = mini_amd.Frontier(gpu_ctx, n), mini_amd.Frontier(gpu_ctx, n)
    fa.load(np.array([src], dtype=np.int32))
    want = oracle.bfs_cpu(ro, ci, src)
    for it in range(64):
        kept = bfs.advance_filter_fused(fa, fb, it)
        ids = fb.read()
        assert kept == len(ids)
        assert np.array_equal(np.sort(ids), np.where(want == it + 1)[0])
        if kept == 0:
            break
        fa, fb = fb, fa
    assert np.array_equal(bfs.labels(), want)


# ---- whole traversals ---------------------------------------------------------------------------
@pytest.mark.parametrize("direct", [1, 0])
@pytest.mark.parametrize("scale", [8, 10, 13, 16])
def test_bfs_rmat_parity_all_paths(gpu_ctx, oracle, rmat_graphs, scale, direct, monkeypatch):
    """direct = 1: small levels chained inside a push launch (include/mgx/bfs_fused_chain.hpp), 0: every level device-wide"""
    import mini_amd
    from mini_amd import rmat
    monkeypatch.setenv("MGX_BFS_CHAIN_MAX_EDGES", "6144" if direct else "0")
    n, ro, ci, w = rmat_graphs[scale]
    g = _graph(gpu_ctx, ro, ci)
    deg = np.diff(ro)
    sources = [int(np.argmax(deg))] + rmat.pick_sources(ro, 3, scale)
    iso = np.where(deg == 0)[0]
    if len(iso):
        sources.append(int(iso[0]))                                    # isolated source: only itself
    bfs = mini_amd.BfsProblem(g, sources[0])
    for src in sources:
        want = oracle.bfs_cpu(ro, ci, src)
        st = bfs.run(src)
        got = bfs.labels()
        assert np.array_equal(got, want), "fused, src=%d" % src
        assert st["reached"] == int((want >= 0).sum())
        assert st["m_t"] == int(deg[want >= 0].sum())
        assert st["levels"] == (int(want.max()) + (1 if deg[want == want.max()].sum() > 0 else 0) if deg[src] else 0)
        trace = bfs.level_trace()
        for lv, (nf, ne) in enumerate(trace):
            on = (want == lv) & (deg > 0)
            assert nf == int(on.sum()) and ne == int(deg[on].sum())
        for alpha in (None, 0.2, 3.0):
            bfs.reset(src)
            bfs.enact_pushpull(alpha)
            assert np.array_equal(bfs.labels(), want), "pushpull alpha=%s src=%d" % (alpha, src)


@pytest.mark.parametrize("scale,hot_min_edges,long_min,small_max",
                         [(10, 0, 64, 0), (10, 0, 64, 8192), (13, 0, 8, 100), (13, 0, 0, 8192), (16, 0, 64, 8192),
                          (16, 65536, 64, 0), (16, 0, 1, 3000), (16, 1 << 30, 16, 8192),
                          (10, 0, 64, -1), (13, 0, 8, -1), (16, 0, 64, -1), (16, 65536, 0, -1), (16, 1 << 30, 1, -1)])
def test_bfs_hub_first_layout_and_lds_hot_bitmap(gpu_ctx, oracle, torch_mod, rmat_graphs, scale, hot_min_edges, long_min,
                                                 small_max, monkeypatch):
    """fused traversal on the degree-sorted layout (hot prefix of the visited snapshot in LDS): labels must
    come back in ORIGINAL ids and equal the oracle's.  hot_min_edges=0 forces the LDS path on small graphs,
    2^30 keeps every probe in L2; long_min moves rows between the row-wise streaming kernel and the
    per-edge-rank kernel (0: no long-row queue, 1: every row is streamed); small_max: levels up to that many
    edges are chained inside block 0 of a push launch (0 / -1: none).  The unit blocks of the long rows are built when
    the layout is attached (with this long_min); small_max 8192 / 3000 also force them on for every level that may use
    them, 0 keeps every level on the queue walk."""
    import mini_amd
    from mini_amd import rmat
    torch = torch_mod
    monkeypatch.setenv("MGX_BFS_HOT_MIN_EDGES", str(hot_min_edges))
    monkeypatch.setenv("MGX_BFS_LONG_MIN", str(long_min))
    monkeypatch.setenv("MGX_BFS_CHAIN_MAX_EDGES", str(max(small_max, 0)))
    monkeypatch.setenv("MGX_BFS_DENSE", {8192: "1000000", 3000: "1000000", 0: "0"}.get(small_max, "16"))
    n, ro, ci, w = rmat_graphs[scale]
    d_ro, d_ci = torch.from_numpy(ro).cuda(), torch.from_numpy(ci).cuda()
    g = mini_amd.Graph.from_device(gpu_ctx, n, len(ci), d_ro, d_ci)
    lro, lci, new_of_old, old_of_new = rmat.degree_order(d_ro, d_ci)
    # the layout is the same graph: degrees sorted descending, maps inverse of each other
    ldeg = np.diff(lro.cpu().numpy())
    assert np.all(np.diff(ldeg) <= 0)
    assert np.array_equal(new_of_old.cpu().numpy()[old_of_new.cpu().numpy()], np.arange(n))
    g.attach_layout(lro, lci, new_of_old, old_of_new)
    deg = np.diff(ro)
    bfs = mini_amd.BfsProblem(g, 0)
    iso = np.where(deg == 0)[0]
    for src in [int(np.argmax(deg))] + rmat.pick_sources(ro, 3, scale + 7) + ([int(iso[0])] if len(iso) else []):
        want = oracle.bfs_cpu(ro, ci, src)
        st = bfs.run(src)
        assert np.array_equal(bfs.labels(), want), "src=%d" % src
        assert st["reached"] == int((want >= 0).sum()) and st["m_t"] == int(deg[want >= 0].sum())
        # the operator path ignores the layout
        bfs.reset(src)
        bfs.enact_pushpull()
        assert np.array_equal(bfs.labels(), want)


@pytest.mark.parametrize("direct", [1, 0])
@pytest.mark.parametrize("scale", [10, 13, 16])
@pytest.mark.parametrize("layout", [False, True])
def test_bfs_direction_optimizing_fused(gpu_ctx, oracle, torch_mod, rmat_graphs, scale, layout, direct, monkeypatch):
    """fused direction-optimising traversal (bottom-up levels once unvisited < frontier*alpha,
    bfs_enactor.hxx:68): labels equal the top-down oracle for every switch point, incl. pull from level 0.
    direct = 0: the top-down levels read their long rows from the unit blocks whenever they may (layout runs)."""
    import mini_amd
    from mini_amd import rmat
    torch = torch_mod
    monkeypatch.setenv("MGX_BFS_HOT_MIN_EDGES", "0")
    monkeypatch.setenv("MGX_BFS_DENSE", "16" if direct else "1000000")
    n, ro, ci, w = rmat_graphs[scale]
    d_ro, d_ci = torch.from_numpy(ro).cuda(), torch.from_numpy(ci).cuda()
    g = mini_amd.Graph.from_device(gpu_ctx, n, len(ci), d_ro, d_ci)
    if layout:
        g.attach_layout(*rmat.degree_order(d_ro, d_ci))
    deg = np.diff(ro)
    bfs = mini_amd.BfsProblem(g, 0)
    for src in [int(np.argmax(deg))] + rmat.pick_sources(ro, 2, scale + 3):
        want = oracle.bfs_cpu(ro, ci, src)
        seen_pull = False
        for alpha in (0.05, 1.0, 4.0, 64.0, 1e9):
            st = bfs.run(src, mode=mini_amd.MGX_BFS_DIRECTION_OPT, alpha=alpha)
            assert np.array_equal(bfs.labels(), want), "alpha=%g src=%d" % (alpha, src)
            assert st["reached"] == int((want >= 0).sum()) and st["m_t"] == int(deg[want >= 0].sum())
            assert st["push_levels"] <= st["levels"]
            seen_pull = seen_pull or st["pull_edges"] > 0
            if alpha == 1e9:
                assert st["push_levels"] == 0 and st["push_edges"] == 0
        assert seen_pull


def test_bfs_direction_optimizing_directed_needs_genuine_csc(gpu_ctx, oracle):
    """directed input: bottom-up levels walk IN-edges, so the graph must carry a genuine CSC
    (the reference's CSC is a CSR copy and its pull phase is wrong there)"""
    import mini_amd
    rng = np.random.default_rng(11)
    n, e = 4000, 30000
    t0 = rng.integers(0, n, size=e).astype(np.int32)
    t1 = rng.integers(0, n, size=e).astype(np.int32)
    ro, ci, _ = oracle.csr_from_tuples(n, t0, t1, None, undir=False)       # row t1 -> neighbour t0
    co, ri, _ = oracle.csr_from_tuples(n, t1, t0, None, undir=False)       # transpose: row t0 <- t1
    g = mini_amd.Graph.from_host(gpu_ctx, ro, ci, None, co, ri)
    bfs = mini_amd.BfsProblem(g, 0)
    for src in (int(t1[0]), int(t1[5])):
        want = oracle.bfs_cpu(ro, ci, src)
        for alpha in (0.5, 8.0, 1e9):
            bfs.run(src, mode=mini_amd.MGX_BFS_DIRECTION_OPT, alpha=alpha)
            assert np.array_equal(bfs.labels(), want), alpha


def test_bfs_directed_graph_with_zero_outdegree_vertices(gpu_ctx, oracle):
    """directed (undir=false) graphs have reachable vertices with no out-edges: they get a label
    but never enter the fused frontier."""
    import mini_amd
    rng = np.random.default_rng(7)
    n, e = 5000, 20000
    t0 = rng.integers(0, n, size=e).astype(np.int32)
    t1 = rng.integers(0, n // 4, size=e).astype(np.int32)              # only a quarter have out-edges
    ro, ci, w = oracle.csr_from_tuples(n, t0, t1, None, undir=False)
    g = _graph(gpu_ctx, ro, ci)
    bfs = mini_amd.BfsProblem(g, 0)
    for src in (int(t1[0]), int(t1[1]), n - 1):
        want = oracle.bfs_cpu(ro, ci, src)
        bfs.run(src)
        assert np.array_equal(bfs.labels(), want)
        bfs.reset(src)
        bfs.enact_pushpull()
        assert np.array_equal(bfs.labels(), want)


@pytest.mark.parametrize("small_max", [8192, 0])
def test_bfs_long_chain_many_levels(gpu_ctx, oracle, monkeypatch, small_max):
    """a path graph: > levels_per_sync levels, frontier of one vertex each.  small_max=8192: all 300 levels run
    inside ONE push launch, chained by its block 0; 0: every level goes through the device-wide kernels, slot by slot."""
    import mini_amd
    monkeypatch.setenv("MGX_BFS_CHAIN_MAX_EDGES", str(small_max))
    n = 300
    t0 = np.arange(0, n - 1, dtype=np.int32)
    t1 = np.arange(1, n, dtype=np.int32)
    ro, ci, w = oracle.csr_from_tuples(n, t0, t1, None, undir=True)
    g = _graph(gpu_ctx, ro, ci)
    bfs = mini_amd.BfsProblem(g, 0)
    st = bfs.run(0)
    assert np.array_equal(bfs.labels(), np.arange(n, dtype=np.int32))
    assert st["levels"] == n          # frontiers at depth 0..n-1 all expand an edge
    assert st["small_levels"] == (n if small_max else 0)
    tr = bfs.level_trace()
    assert len(tr) == n and all(t == (1, 2) for t in tr[1:-1]) and tr[0] == (1, 1) and tr[-1] == (1, 1)
    st = bfs.run(n - 1)
    assert np.array_equal(bfs.labels(), np.arange(n - 1, -1, -1, dtype=np.int32))
    assert st["levels"] == n
    assert st["small_levels"] == (n if small_max else 0)
    if small_max:
        assert st["slots"] <= 3       # launched: an M launch, ONE device-wide slot, an M launch -- the chain ran the traversal
    bfs.run(n // 2)                   # 150 levels
    assert np.array_equal(bfs.labels(), np.abs(np.arange(n) - n // 2).astype(np.int32))


@pytest.mark.parametrize("small_max", [256, 0, 1 << 20, -1])
@pytest.mark.parametrize("long_min", [64, 1])
def test_bfs_long_row_queue_padding_boundaries(gpu_ctx, oracle, monkeypatch, small_max, long_min):
    """The long-row queue counts degrees rounded up to 64 and keeps degree & 63 in the low bits of its offsets
    (bfs_lq_* in include/mgx/bfs_fused.hpp).  A tree whose second level has rows of every length around the
    multiples of 64 (and a 5000-edge row that spans several slices), expanded by the stream kernel (small_max 0 /
    256 / -1), by the chain of small levels where a level fits (small_max 2^20: capped at BFS_CHAIN_CAP) and with every
    row in the long queue (long_min 1): labels, levels and the traversed-edge count (true edges, not padded ones) must
    equal the oracle's.  Also through the unit blocks of a library-built layout (forced on)."""
    import mini_amd
    monkeypatch.setenv("MGX_BFS_CHAIN_MAX_EDGES", str(max(small_max, 0)))
    monkeypatch.setenv("MGX_BFS_DENSE", "1000000")
    monkeypatch.setenv("MGX_BFS_LONG_MIN", str(long_min))
    monkeypatch.setenv("MGX_BFS_HOT_MIN_EDGES", "0")
    degs = [1, 2, 62, 63, 64, 65, 66, 126, 127, 128, 129, 190, 191, 192, 193, 255, 256, 257, 511, 512, 513, 5000]
    t0, t1 = [], []
    nxt = 1 + len(degs)
    for i, d in enumerate(degs):
        hub = 1 + i
        t0.append(0); t1.append(hub)                 # source -> hub
        for _ in range(d - 1):                       # the hub's other d - 1 neighbours (its degree is d with the source)
            t0.append(hub); t1.append(nxt); nxt += 1
    n = nxt
    ro, ci, w = oracle.csr_from_tuples(n, np.array(t0, dtype=np.int32), np.array(t1, dtype=np.int32), None, undir=True)
    deg = np.diff(ro)
    assert sorted(deg[1:1 + len(degs)].tolist()) == sorted(degs)
    g = _graph(gpu_ctx, ro, ci)
    if small_max in (0, -1):
        g.build_layout()                             # + unit blocks, read by every level whose frontier bitmap is current
    bfs = mini_amd.BfsProblem(g, 0)
    for src in (0, 5, len(degs), n - 1):             # the root, a 64-edge hub, the 5000-edge hub, a leaf
        want = oracle.bfs_cpu(ro, ci, src)
        st = bfs.run(src)
        assert np.array_equal(bfs.labels(), want), src
        if small_max in (0, -1) and src == 0:
            assert st["dense_slots"] >= 1            # the hubs' level: 22 long rows read from the unit blocks
        reached = want >= 0
        assert st["reached"] == int(reached.sum())
        assert st["m_t"] == int(deg[reached].sum())
        assert st["levels"] == int(want.max()) + 1
        tr = bfs.level_trace()
        for lv, (nf, ne) in enumerate(tr):
            at = want == lv
            assert nf == int((at & (deg > 0)).sum()) and ne == int(deg[at].sum()), (src, lv)


@pytest.mark.parametrize("scale", [8, 13, 16])
def test_library_built_layout_equals_torch_construction(gpu_ctx, oracle, torch_mod, rmat_graphs, scale):
    """mgx_graph_build_layout (device-side degree sort + renumbering, rocPRIM) against mini_amd.rmat.degree_order
    (torch ops): identical id maps, offsets and neighbour lists; weights travel with their edges (compared as the
    multiset of (neighbour, weight) per row: equal neighbours may come in either order); traversals on the built
    layout equal the oracle."""
    import mini_amd
    from mini_amd import rmat
    torch = torch_mod
    n, ro, ci, w = rmat_graphs[scale]
    d_ro, d_ci, d_w = torch.from_numpy(ro).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(w).cuda()
    g = mini_amd.Graph.from_device(gpu_ctx, n, len(ci), d_ro, d_ci, d_w)
    g.build_layout(weights=True)
    lro, lci, n2o, o2n, lw = g.layout_arrays(weights=True)
    t_lro, t_lci, t_n2o, t_o2n, t_lw = [t.cpu().numpy() for t in rmat.degree_order(d_ro, d_ci, d_w)]
    assert np.array_equal(o2n, t_o2n) and np.array_equal(n2o, t_n2o)
    assert np.array_equal(lro, t_lro) and np.array_equal(lci, t_lci)
    key = lambda nb, ww: np.lexsort((ww, nb, np.repeat(np.arange(n), np.diff(lro))))
    a, b = key(lci, lw), key(t_lci, t_lw)
    assert np.array_equal(lci[a], t_lci[b]) and np.array_equal(lw[a], t_lw[b])
    deg = np.diff(ro)
    bfs = mini_amd.BfsProblem(g, 0)
    sssp = mini_amd.SsspProblem(g, 0)
    for src in [int(np.argmax(deg))] + rmat.pick_sources(ro, 2, scale + 11):
        bfs.run(src)
        assert np.array_equal(bfs.labels(), oracle.bfs_cpu(ro, ci, src))
        sssp.run(src)
        want, _, _ = oracle.sssp_enact(ro, ci, w, src, 8.0)
        assert np.array_equal(sssp.distances(), want)
    # a graph wrapped without weights has unit weights (graph.hxx:126's default): its layout carries those
    g1 = mini_amd.Graph.from_device(gpu_ctx, n, len(ci), d_ro, d_ci).build_layout(weights=True)
    assert np.all(g1.layout_arrays(weights=True)[4] == 1.0)


@pytest.mark.parametrize("direct", [1, 0])
def test_degenerate_graphs_through_the_fused_loops(gpu_ctx, oracle, direct, monkeypatch):
    """one vertex without edges, a single edge, a self-loop, 100 isolated vertices, 33 vertices in a ring: fused BFS
    (push and direction-optimising), fused SSSP, with and without the library-built hub-first copy"""
    import mini_amd
    monkeypatch.setenv("MGX_BFS_CHAIN_MAX_EDGES", "6144" if direct else "0")
    monkeypatch.setenv("MGX_BFS_DENSE", "16" if direct else "1000000")
    cases = [(1, [], []), (2, [0], [1]), (3, [1], [1]), (100, [], []), (33, list(range(33)), [(i + 1) % 33 for i in range(33)])]
    for n, t0, t1 in cases:
        wv = (np.arange(len(t0)) % 7).astype(np.float32)
        ro, ci, w = oracle.csr_from_tuples(n, np.array(t0, dtype=np.int32), np.array(t1, dtype=np.int32), wv, undir=True)
        for layout in (False, True):
            g = _graph(gpu_ctx, ro, ci, w)
            if layout:
                g.build_layout(weights=True)
            bfs, sssp = mini_amd.BfsProblem(g, 0), mini_amd.SsspProblem(g, 0)
            for src in sorted({0, n - 1, n // 2}):
                want = oracle.bfs_cpu(ro, ci, src)
                st = bfs.run(src)
                assert np.array_equal(bfs.labels(), want), (n, src, layout)
                assert st["reached"] == int((want >= 0).sum())
                bfs.run(src, mode=mini_amd.MGX_BFS_DIRECTION_OPT, alpha=2.0)
                assert np.array_equal(bfs.labels(), want), (n, src, layout, "do")
                dist, _, _ = oracle.sssp_enact(ro, ci, w, src, 8.0)
                sssp.run(src)
                assert np.array_equal(sssp.distances(), dist), (n, src, layout, "sssp")


@pytest.mark.parametrize("cold", [1, 0])
@pytest.mark.parametrize("direct", [1, 0])
def test_bfs_far_hub_is_not_discovered_early(gpu_ctx, oracle, monkeypatch, cold, direct):
    """The biggest hub (layout vertex 0) sits six levels away from the source, while a medium hub next to the source
    gives the stream kernel long rows to read from level 1 on.  Guards the placeholders of the stream kernel's
    software pipeline (vertex id 0): the cold-test variant once tested them for real and discovered layout vertex 0
    at the first level that had a long row (found by tools/fuzz_parity.py with MGX_BFS_COLD_TEST=1)."""
    import mini_amd
    monkeypatch.setenv("MGX_BFS_COLD_TEST", str(cold))
    monkeypatch.setenv("MGX_BFS_HOT_MIN_EDGES", "0")
    monkeypatch.setenv("MGX_BFS_CHAIN_MAX_EDGES", "6144" if direct else "0")
    monkeypatch.setenv("MGX_BFS_DENSE", "0" if direct else "1000000")
    t0, t1 = [], []
    nxt = 8
    for k in range(300):            # medium hub 1 next to the source 0
        t0.append(1); t1.append(nxt); nxt += 1
    t0.append(0); t1.append(1)
    for a, b in ((0, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7)):      # a path from the source to the big hub 7
        t0.append(a); t1.append(b)
    for k in range(5000):
        t0.append(7); t1.append(nxt); nxt += 1
    n = nxt
    ro, ci, w = oracle.csr_from_tuples(n, np.array(t0, dtype=np.int32), np.array(t1, dtype=np.int32), None, undir=True)
    g = _graph(gpu_ctx, ro, ci).build_layout()
    bfs = mini_amd.BfsProblem(g, 0)
    for src in (0, 1, 8, n - 1):
        want = oracle.bfs_cpu(ro, ci, src)
        bfs.run(src)
        assert np.array_equal(bfs.labels(), want), (src, cold, direct)


VARIANTS = [{}, {"MGX_BFS_COLD_TEST": "1"}, {"MGX_BFS_COLD_TEST": "1", "MGX_BFS_HOT_MIN_EDGES": "0"},
            {"MGX_BFS_LONG_MIN": "1", "MGX_BFS_COLD_TEST": "1"}, {"MGX_BFS_LONG_MIN": "0"},
            {"MGX_BFS_HOT_MIN_EDGES": "1000000000", "MGX_BFS_LONG_MIN": "8"}, {"MGX_BFS_MERGED_PUSH": "0", "MGX_BFS_COLD_TEST": "1"},
            {"MGX_BFS_CHAIN_MAX_EDGES": "0"}, {"MGX_BFS_CHAIN_MAX_EDGES": "64", "MGX_BFS_DENSE": "1000000", "MGX_BFS_HOT_MIN_EDGES": "0"},
            {"MGX_BFS_DENSE": "1000000"}, {"MGX_BFS_DENSE": "0", "MGX_BFS_CHAIN_MAX_EDGES": "100000"},
            {"MGX_BFS_DENSE": "1000000", "MGX_BFS_LONG_MIN": "1"}, {"MGX_BFS_DENSE": "1000000", "MGX_BFS_MERGED_PUSH": "0"},
            {"MGX_BFS_VSHORT": "1000000"}, {"MGX_BFS_VSHORT": "1000000", "MGX_BFS_LONG_MIN": "8", "MGX_BFS_HOT_MIN_EDGES": "0"},
            {"MGX_BFS_VSHORT": "1000000", "MGX_BFS_DENSE": "1000000", "MGX_BFS_DEFER": "1", "MGX_BFS_CHAIN_MAX_EDGES": "0", "MGX_BFS_HOT_MIN_EDGES": "0"},
            {"MGX_BFS_VSHORT": "0", "MGX_BFS_DENSE": "0"},
            {"MGX_BFS_BUILD_LIST": "1"}, {"MGX_BFS_DEFER": "0"}, {"MGX_BFS_DEFER": "1", "MGX_BFS_HOT_MIN_EDGES": "0"},
            {"MGX_BFS_DEFER": "1", "MGX_BFS_HOT_MIN_EDGES": "0", "MGX_BFS_BUILD_LIST": "1", "MGX_BFS_DENSE": "1000000"},
            # lazy queues (bfs_build_is_lazy) behind EVERY device-wide level / never
            {"MGX_BFS_LAZY": "1048576"}, {"MGX_BFS_LAZY": "1048576", "MGX_BFS_CHAIN_MAX_EDGES": "0", "MGX_BFS_HOT_MIN_EDGES": "0"},
            {"MGX_BFS_LAZY": "1048576", "MGX_BFS_DEFER": "1", "MGX_BFS_LONG_MIN": "8"}, {"MGX_BFS_LAZY": "0"},
            {"MGX_BFS_LAZY": "1048576", "MGX_BFS_MERGED_PUSH": "0"},
            # the chain of small levels at the start: inside slot 0's push launch instead of a launch of its own
            {"MGX_BFS_SEED_CHAIN": "0"}, {"MGX_BFS_SEED_CHAIN": "0", "MGX_BFS_DENSE": "1000000", "MGX_BFS_LAZY": "1048576"},
            {"MGX_BFS_MINI": "2"}, {"MGX_BFS_MINI": "0", "MGX_BFS_TAIL_FRONT": "0"}, {"MGX_BFS_TAIL_CHAIN": "0"}, {"MGX_BFS_CHAIN_BIG_EDGES": "100"}, {"MGX_BFS_CHAIN_MAX_EDGES": "64", "MGX_BFS_CHAIN_BIG_EDGES": "12288", "MGX_BFS_LAZY": "1048576"},
            # round 4: the unit blocks' 32-bit entries instead of the 24-bit copy (every level on them / by the default rule), the deferred
            # range shortened to one run and to a third, the per-source launch plan off / on with M launches forced on small graphs
            {"MGX_BFS_PACK24": "0", "MGX_BFS_DENSE": "1000000"}, {"MGX_BFS_PACK24": "0"}, {"MGX_BFS_DENSE": "1000000", "MGX_BFS_DEFER": "1", "MGX_BFS_DEFER_WORDS": "32"},
            {"MGX_BFS_DEFER_WORDS": "6016", "MGX_BFS_DEFER": "1", "MGX_BFS_HOT_MIN_EDGES": "0"}, {"MGX_BFS_DEFER_WORDS": "0"},
            {"MGX_BFS_MINI": "2", "MGX_BFS_SRC_PLAN": "0"}, {"MGX_BFS_MINI": "2", "MGX_BFS_SRC_PLAN": "1", "MGX_BFS_CHAIN_BIG_EDGES": "64"}]


@pytest.mark.parametrize("variant", range(len(VARIANTS)))
def test_bfs_kernel_variants_on_random_graphs(gpu_ctx, oracle, monkeypatch, variant):
    """A small randomised campaign (tools/fuzz_parity.py is the long one) with the kernel variants forced that the
    default thresholds only pick on big or unusual inputs: cold-test instances, every / no row in the long-row queue,
    no LDS prefix, unmerged launches, no chains of small levels / short ones, the unit blocks forced on for every level
    that may use them (also with every row in them) / off."""
    import mini_amd
    from tests.conftest import skip_unless_lab
    skip_unless_lab(VARIANTS[variant])      # (shapes that lost their A/B runs live in the lab library only)
    for k, v in VARIANTS[variant].items():
        monkeypatch.setenv(k, v)
    rng = np.random.default_rng(100 + variant)
    for trial in range(6):
        kind = trial % 3
        if kind == 0:      # star forest, hubs of every size
            h, n = int(rng.integers(1, 200)), 30000
            deg = rng.integers(1, 2000, size=h)
            t0 = np.repeat(np.arange(h), deg).astype(np.int32)
            t1 = rng.integers(h, n, size=int(deg.sum())).astype(np.int32)
        elif kind == 1:    # uniform random
            n = int(rng.integers(2, 50000)); e = int(rng.integers(1, 6 * n))
            t0 = rng.integers(0, n, size=e).astype(np.int32); t1 = rng.integers(0, n, size=e).astype(np.int32)
        else:              # R-MAT
            n, ro, ci, w = oracle.rmat_csr(int(rng.integers(8, 15)), int(rng.integers(2, 20)), int(rng.integers(1, 1 << 20)))
        if kind != 2:
            ro, ci, w = oracle.csr_from_tuples(n, t0, t1, None, undir=bool(trial % 2 == 0) or kind == 0)
        g = _graph(gpu_ctx, ro, ci)
        if trial % 2 == 0:
            g.build_layout()
        deg = np.diff(ro)
        bfs = mini_amd.BfsProblem(g, 0)
        for src in [int(np.argmax(deg))] + [int(x) for x in rng.integers(0, n, size=3)]:
            want = oracle.bfs_cpu(ro, ci, src)
            st = bfs.run(src)
            assert np.array_equal(bfs.labels(), want), (variant, trial, src)
            assert st["m_t"] == int(deg[want >= 0].sum())


@pytest.mark.parametrize("build_list", [0, 1])
@pytest.mark.parametrize("hot_min_edges", [0, 1 << 30])
def test_sssp_fused_with_and_without_lds_distance_bounds(gpu_ctx, oracle, monkeypatch, hot_min_edges, build_list):
    """the bfloat16 upper bounds of the hubs' distances (sssp_fused.hpp) forced on for every iteration, and off; the direct
    queue build (k_sssp_build2) and the list-based one"""
    import mini_amd
    monkeypatch.setenv("MGX_SSSP_HOT_MIN_EDGES", str(hot_min_edges))
    monkeypatch.setenv("MGX_SSSP_BUILD_LIST", str(build_list))
    rng = np.random.default_rng(5 + (hot_min_edges > 0))
    for trial in range(4):
        n, ro, ci, _ = oracle.rmat_csr(int(rng.integers(8, 15)), int(rng.integers(2, 20)), int(rng.integers(1, 1 << 20)))
        w = (rng.random(len(ci)) * (64.0 if trial % 2 else 1.0)).astype(np.float32)
        if trial % 2:
            w = np.floor(w)
        g = _graph(gpu_ctx, ro, ci, w)
        if trial < 3:
            g.build_layout(weights=True)
        sssp = mini_amd.SsspProblem(g, 0)
        deg = np.diff(ro)
        for src in [int(np.argmax(deg))] + [int(x) for x in rng.integers(0, n, size=2)]:
            want, _, _ = oracle.sssp_enact(ro, ci, w, src, 8.0)
            sssp.run(src)
            assert np.array_equal(sssp.distances(), want), (trial, src)


def test_sssp_fused_float_weights_and_big_frontiers(gpu_ctx, oracle, rmat_graphs):
    """fused SSSP loop on RMAT-16 (frontiers of several thousand marked vertices per workgroup: the queue build
    runs more than one batch) with NON-integer weights: the min-plus fixed point is unique, so distances are
    bit-identical to Dijkstra's in float32 (oracle.sssp_dijkstra_f32)."""
    import mini_amd
    from mini_amd import rmat
    n, ro, ci, w = rmat_graphs[16]
    rng = np.random.default_rng(7)
    wf = (rng.random(len(ci), dtype=np.float32) * 9.0 + 0.125).astype(np.float32)
    g = _graph(gpu_ctx, ro, ci, wf)
    sssp = mini_amd.SsspProblem(g, 0)
    for src in [int(np.argmax(np.diff(ro)))] + rmat.pick_sources(ro, 2, 1234):
        st = sssp.run(src)
        want = oracle.sssp_dijkstra_f32(ro, ci, wf, src)
        got = sssp.distances()
        fin = want < FLT_MAX
        assert np.array_equal(got[fin], want[fin]) and np.all(got[~fin] == FLT_MAX), "src=%d" % src
        assert st["iterations"] >= 1 and st["relaxations"] >= int(np.diff(ro)[src])


@pytest.mark.parametrize("scale", [8, 10, 13])
def test_sssp_rmat_parity(gpu_ctx, oracle, rmat_graphs, scale):
    import mini_amd
    from mini_amd import rmat
    n, ro, ci, w = rmat_graphs[scale]
    g = _graph(gpu_ctx, ro, ci, w)
    sssp = mini_amd.SsspProblem(g, 0)
    for src in [int(np.argmax(np.diff(ro)))] + rmat.pick_sources(ro, 2, scale + 100):
        sssp.reset(src)
        st = sssp.enact(1.5)
        want, _, ost = oracle.sssp_enact(ro, ci, w, src, 1.5)
        dist = sssp.distances()
        assert np.array_equal(dist, want), "src=%d" % src                # bit-exact (<= 1e-6 rel required)
        preds = sssp.preds()
        # preds are racy upstream; what must hold: pred is -1 exactly for src/unreached,
        # and is an in-neighbour otherwise
        fin = dist < FLT_MAX
        assert np.all(preds[~fin] == -1)
        for v in np.where(fin)[0][:300]:
            if v == src:
                continue
            p = preds[v]
            assert p >= 0 and v in ci[ro[p]:ro[p + 1]]
        st2 = sssp.run(src)
        assert np.array_equal(sssp.distances(), want)
        assert st2["iterations"] >= 1
    # the fused loop in layout space (hub-first ids, weights permuted with the edges): distances come back in
    # original ids and equal the oracle's
    import torch
    d_ro, d_ci, d_w = torch.from_numpy(ro).cuda(), torch.from_numpy(ci).cuda(), torch.from_numpy(w).cuda()
    g.attach_layout(*rmat.degree_order(d_ro, d_ci, d_w))
    for src in [int(np.argmax(np.diff(ro)))] + rmat.pick_sources(ro, 2, scale + 100):
        want, _, _ = oracle.sssp_enact(ro, ci, w, src, 1.5)
        sssp.run(src)
        assert np.array_equal(sssp.distances(), want), "layout src=%d" % src


@pytest.mark.parametrize("scale,layout", [(10, False), (13, False), (16, False), (14, True), (16, True)])
def test_pr_matches_oracle(gpu_ctx, oracle, rmat_graphs, scale, layout):
    """pr_enactor.hxx:41-79 on the neighbour-reduce operator against the oracle's serial restatement -- R-MAT 10 .. 16 (65 536
    vertices, 2 M entries: hub rows that span hundreds of tiles), on the CSR as loaded and, where the full-frontier path of
    mgx/nreduce.hpp applies, on a graph that carries the hub-first layout with its unit blocks"""
    import mini_amd
    n, ro, ci, w = rmat_graphs[scale] if scale in rmat_graphs else oracle.rmat_csr(scale, 16, 100 + scale)
    g = _graph(gpu_ctx, ro, ci)
    if layout:
        g.build_layout()
    for iters in (1, 3):
        pr = mini_amd.PrProblem(g, iters)
        lens = pr.enact()
        want, wlens = oracle.pr_enact(ro, ci, iters)
        got = pr.ranks()
        # float sums: segment order differs between the serial oracle (left to right, one float accumulator) and the tiled reduce
        # (lanes, then a shuffle tree): the gap grows with the longest row -- measured 1.0e-5 relative at R-MAT 10 (2 105 entries),
        # 4.0e-5 at 13 (7 370), 1.7e-4 at 16 (26 125) -- so the tolerance is half a float ulp per term of the longest row, with the
        # north star's 2e-5 as the floor where rows are short
        rtol = max(2e-5, 0.5 * 6e-8 * float(np.diff(ro).max()))
        assert np.allclose(got, want, rtol=rtol, atol=1e-6), (scale, iters, float(np.abs(got - want).max()))
        assert abs(float(got.sum()) - float(want.sum())) <= 2e-5 * float(want.sum())
        assert len(lens) == len(wlens)
        if iters == 1:
            assert lens[0] == wlens[0]


# ---- error behaviour ------------------------------------------------------------------------------
def test_frontier_overflow_and_bad_arguments_are_statuses(gpu_ctx, oracle, rmat_graphs):
    import mini_amd
    n, ro, ci, w = rmat_graphs[8]
    g = _graph(gpu_ctx, ro, ci, w)
    f = mini_amd.Frontier(gpu_ctx, 4)
    with pytest.raises(mini_amd.MgxError) as e:
        f.load(np.arange(5, dtype=np.int32))
    assert e.value.status == mini_amd.MGX_E_FRONTIER_OVERFLOW
    with pytest.raises(mini_amd.MgxError):
        mini_amd.BfsProblem(g, n)                   # src out of range
    # advance into a too-small output frontier: reference exit(0)s, we return the status
    src = int(np.argmax(np.diff(ro)))
    bfs = mini_amd.BfsProblem(g, src)
    fin = mini_amd.Frontier(gpu_ctx, 4).load(np.array([src], dtype=np.int32))
    with pytest.raises(mini_amd.MgxError) as e:
        bfs.advance(fin, f, 0)
    assert e.value.status == mini_amd.MGX_E_FRONTIER_OVERFLOW
    wneg = w.copy()
    wneg[3] = -1.0
    gneg = _graph(gpu_ctx, ro, ci, wneg)
    with pytest.raises(mini_amd.MgxError) as e:
        mini_amd.SsspProblem(gneg, 0)
    assert e.value.status == mini_amd.MGX_E_NEGATIVE_WEIGHT


# ---- size-independent properties at a large size ---------------------------------------------------
def test_bfs_large_rmat_properties(gpu_ctx, torch_mod):
    """RMAT scale 20 on the device (33.5 M CSR entries): BFS-tree validity checked with device ops
    (no oracle at this size): label[src]=0; |label[u]-label[v]|<=1 over every edge with both ends
    reached; an edge never leaves the reached set; every reached v != src has a neighbour one
    level closer; the fused path and the operator path agree bit for bit."""
    import mini_amd
    from mini_amd import rmat
    torch = torch_mod
    g = rmat.rmat_csr(gpu_ctx, scale=20, edgefactor=16, seed=20)
    graph = mini_amd.Graph.from_device(gpu_ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"])
    ro = g["row_offsets"].to(torch.int64)
    deg = ro[1:] - ro[:-1]
    rows = torch.repeat_interleave(torch.arange(g["n"], device="cuda"), deg)
    cols = g["col_indices"].to(torch.int64)
    src = rmat.pick_sources(g["row_offsets"].cpu().numpy(), 1, 20)[0]
    bfs = mini_amd.BfsProblem(graph, src)
    st = bfs.run(src)
    lab = torch.from_numpy(bfs.labels()).cuda().to(torch.int64)
    assert lab[src] == 0
    lr, lc = lab[rows], lab[cols]
    assert bool(((lr >= 0) == (lc >= 0)).all())
    both = lr >= 0
    assert int((lr[both] - lc[both]).abs().max()) <= 1
    big = torch.full((g["n"],), 1 << 40, dtype=torch.int64, device="cuda")
    mn = big.scatter_reduce(0, rows[both], lc[both], reduce="amin", include_self=True)
    reached = lab >= 0
    chk = reached.clone()
    chk[src] = False
    assert bool((mn[chk] == lab[chk] - 1).all())
    assert st["reached"] == int(reached.sum()) and st["m_t"] == int(deg[reached].sum())
    fused = bfs.labels().copy()
    bfs.reset(src)
    bfs.enact_pushpull()
    assert np.array_equal(bfs.labels(), fused)
    # hub-first layout + LDS hot bitmap kernel: same labels, original ids
    lay = rmat.degree_order(g["row_offsets"], g["col_indices"])
    graph2 = mini_amd.Graph.from_device(gpu_ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"]).attach_layout(*lay)
    bfs2 = mini_amd.BfsProblem(graph2, src)
    st2 = bfs2.run(src)
    assert np.array_equal(bfs2.labels(), fused)
    assert st2["m_t"] == st["m_t"] and st2["reached"] == st["reached"]


@pytest.mark.parametrize("layout", [False, True])
def test_sssp_fused_near_far_buckets(gpu_ctx, oracle, layout):
    """delta-stepping in its near / far form (mgx_sssp_run_delta): the same distances as the plain loop and the oracle
    for every bucket width -- tiny (every improvement waits for its bucket), around the mean weight, huge (one bucket ==
    plain Bellman-Ford) -- on integer and on real-valued weights, with and without the hub-first copy; and fewer edge
    relaxations than the plain loop at a sensible width"""
    import mini_amd
    rng = np.random.default_rng(77)
    fewer = 0
    for trial in range(4):
        n, ro, ci, w = oracle.rmat_csr(int(rng.integers(9, 15)), int(rng.integers(4, 20)), int(rng.integers(1, 1 << 20)))
        if trial % 2:
            w = (rng.random(len(ci)) * 10.0).astype(np.float32)
        g = _graph(gpu_ctx, ro, ci, w)
        if layout:
            g.build_layout(weights=True)
        deg = np.diff(ro)
        sssp = mini_amd.SsspProblem(g, 0)
        for src in (int(np.argmax(deg)), int(np.where(deg > 0)[0][3])):
            want = oracle.sssp_dijkstra_f32(ro, ci, w, src)
            base = sssp.run(src)
            assert np.array_equal(sssp.distances(), want)
            for delta in (0.0, 0.5, 4.0, 32.0, 1e9):
                st = sssp.run(src, delta=delta)
                assert np.array_equal(sssp.distances(), want), (trial, src, delta)
                if delta == 4.0 and st["relaxations"] < base["relaxations"]:
                    fewer += 1
    assert fewer >= 4


def test_sssp_near_far_tiny_delta_terminates(gpu_ctx, oracle):
    """with far_min / delta near 2^23 the next threshold (floor(far_min / delta) + 1) * delta can round to
    <= far_min in float32; nothing became near and the loop never ended.  The threshold now always admits the smallest
    waiting distance: weights scaled so that distances are ~1000, delta 1e-4 and 1e-5 (ratios 1e7 .. 1e8), on a graph
    small enough that one bucket per distinct distance still finishes quickly; distances equal float32 Dijkstra's."""
    import mini_amd
    rng = np.random.default_rng(5)
    n, ro, ci, w = oracle.rmat_csr(8, 6, 99)
    w = (rng.random(len(ci)) * 300.0 + 200.0).astype(np.float32)
    g = _graph(gpu_ctx, ro, ci, w)
    deg = np.diff(ro)
    src = int(np.argmax(deg))
    want = oracle.sssp_dijkstra_f32(ro, ci, w, src)
    sssp = mini_amd.SsspProblem(g, src)
    for delta in (1e-4, 1e-5):
        sssp.run(src, delta=delta)
        assert np.array_equal(sssp.distances(), want), delta


@pytest.mark.parametrize("layout", [False, True])
def test_bfs_run_many_equals_one_call_per_source(gpu_ctx, oracle, layout):
    """mgx_bfs_run_many: K traversals enqueued back to back with one host wait -- every traversal's counters equal those
    of a call of its own, the labels left behind are the LAST source's (checked against the oracle for every position
    of the batch by rotating it), direction-optimising batches too; hubs, ordinary and isolated sources mixed"""
    import mini_amd
    rng = np.random.default_rng(31)
    for scale, ef in ((10, 8), (14, 16), (16, 8)):
        n, ro, ci, w = oracle.rmat_csr(scale, ef, 40 + scale)
        g = _graph(gpu_ctx, ro, ci)
        if layout:
            g.build_layout()
        deg = np.diff(ro)
        srcs = [int(np.argmax(deg))] + [int(v) for v in rng.choice(np.where(deg > 0)[0], size=5, replace=False)] + [int(np.where(deg == 0)[0][0])]
        bfs = mini_amd.BfsProblem(g, srcs[0])
        solo = [bfs.run(s) for s in srcs]
        for mode, alpha in ((mini_amd.MGX_BFS_PUSH, 0.0), (mini_amd.MGX_BFS_DIRECTION_OPT, 4.0)):
            for rot in range(len(srcs)):
                batch = srcs[rot:] + srcs[:rot]
                sts, reruns = bfs.run_many(batch, mode, alpha)
                want = oracle.bfs_cpu(ro, ci, batch[-1])
                assert np.array_equal(bfs.labels(), want), (scale, mode, rot)
                for s, st in zip(batch, sts):
                    ref = solo[srcs.index(s)]
                    assert (st["m_t"], st["reached"], st["levels"]) == (ref["m_t"], ref["reached"], ref["levels"]), (scale, mode, rot, s)
        sts, _ = bfs.run_many([], mini_amd.MGX_BFS_PUSH, 0.0)
        assert sts == []


def test_bfs_run_many_longer_than_one_chunk(gpu_ctx, oracle):
    """a long source list is submitted in chunks of 512 traversals (one pinned block of heads, one host wait
    each): 1100 sources = three chunks; every traversal's counters and the last source's labels as with one call per source"""
    import mini_amd
    n, ro, ci, w = oracle.rmat_csr(11, 8, 77)
    g = _graph(gpu_ctx, ro, ci)
    g.build_layout()
    rng = np.random.default_rng(3)
    uniq = [int(v) for v in rng.choice(n, size=40, replace=False)]
    bfs = mini_amd.BfsProblem(g, uniq[0])
    solo = {s: bfs.run(s) for s in uniq}
    batch = [uniq[int(i)] for i in rng.integers(0, len(uniq), size=1100)]
    sts, _ = bfs.run_many(batch, mini_amd.MGX_BFS_PUSH, 0.0)
    assert len(sts) == len(batch)
    for s, st in zip(batch, sts):
        assert (st["m_t"], st["reached"], st["levels"]) == (solo[s]["m_t"], solo[s]["reached"], solo[s]["levels"]), s
    assert np.array_equal(bfs.labels(), oracle.bfs_cpu(ro, ci, batch[-1]))


def test_bfs_run_many_reruns_a_traversal_that_needs_more_slots(gpu_ctx, oracle, monkeypatch):
    """a batch is sized by the launch slots the previous traversals needed: a source whose traversal has many more big
    levels (here: the end of a 40-vertex path next to a star of diameter 2, chains of small levels switched off so that every
    level takes a slot) does not finish inside the batch, is run again on its own and reported in `reruns` -- counters and
    labels are right either way"""
    import mini_amd
    monkeypatch.setenv("MGX_BFS_CHAIN_MAX_EDGES", "0")
    monkeypatch.setenv("MGX_BFS_SEED_CHAIN", "0")
    t0, t1 = [], []
    hub, nleaf = 0, 2000
    for k in range(1, nleaf + 1):                       # a star: depth 1 from the hub, 2 from a leaf
        t0.append(hub); t1.append(k)
    base = nleaf + 1
    length = 40                                         # a path of 40 vertices, a component of its own
    for k in range(length - 1):
        t0.append(base + k); t1.append(base + k + 1)
    n = base + length
    ro, ci, _ = oracle.csr_from_tuples(n, np.array(t0, dtype=np.int32), np.array(t1, dtype=np.int32), None, undir=True)
    g = _graph(gpu_ctx, ro, ci)
    bfs = mini_amd.BfsProblem(g, hub)
    far = n - 1                                         # the end of the path: the deepest traversal of the graph
    ref_far = bfs.run(far)
    # make the handle forget it: four shallow traversals of ANOTHER graph shape are not available here, so size the hint by
    # hand through a fresh handle that has only seen the hub
    bfs = mini_amd.BfsProblem(g, hub)
    ref_hub = None
    for _ in range(4):
        ref_hub = bfs.run(hub)
    assert ref_far["levels"] > ref_hub["levels"] + 8
    sts, reruns = bfs.run_many([hub, far, hub], mini_amd.MGX_BFS_PUSH, 0.0)
    assert reruns >= 1
    assert np.array_equal(bfs.labels(), oracle.bfs_cpu(ro, ci, hub))
    for st, ref in zip(sts, [ref_hub, ref_far, ref_hub]):
        assert (st["m_t"], st["reached"], st["levels"]) == (ref["m_t"], ref["reached"], ref["levels"]), (st, ref, reruns)
    sts, reruns2 = bfs.run_many([far, far], mini_amd.MGX_BFS_PUSH, 0.0)
    assert np.array_equal(bfs.labels(), oracle.bfs_cpu(ro, ci, far))
    assert (sts[1]["m_t"], sts[1]["levels"]) == (ref_far["m_t"], ref_far["levels"])


def test_bfs_per_source_launch_plan(gpu_ctx, oracle, monkeypatch):
    """round 4: the launch sequence of a traversal is chosen per source from graph_device_t::src_shapes (exact shapes of levels
    0 and 1): sources whose first non-chained level is mid-size (the M launch in front absorbs it), sources whose level 1 is too
    big for it (that launch is not enqueued), sources whose levels 0 and 1 both run in the chain -- single calls and batches,
    every order, labels and counters against the oracle and against the plan switched off; no traversal is run twice once
    every class has been seen"""
    import mini_amd
    monkeypatch.setenv("MGX_BFS_MINI", "2")          # (small graphs get no M launches by default)
    rng = np.random.default_rng(77)
    # hubs of 3 000 .. 5 000 entries (their neighbours' level is far above an M launch's 32 768 early edges for some sources,
    # below it for others), a long tail of leaves, isolated vertices
    h, n = 40, 90000
    deg = rng.integers(3000, 5000, size=h)
    t0 = np.concatenate([np.repeat(np.arange(h), deg), rng.integers(h, n - 500, size=60000)]).astype(np.int32)
    t1 = np.concatenate([rng.integers(h, n - 500, size=int(deg.sum())), rng.integers(h, n - 500, size=60000)]).astype(np.int32)
    ro, ci, _ = oracle.csr_from_tuples(n, t0, t1, None, undir=True)
    d = np.diff(ro)
    leaves = np.where((d > 0) & (d <= 3))[0]
    mids = np.where((d >= 20) & (d < 64))[0]
    srcs = [0, 7, int(leaves[0]), int(leaves[len(leaves) // 2]), int(leaves[-1])] + [int(v) for v in mids[:3]] + [n - 1]
    results = {}
    for plan in ("1", "0"):
        monkeypatch.setenv("MGX_BFS_SRC_PLAN", plan)
        g = _graph(gpu_ctx, ro, ci)
        g.build_layout()
        bfs = mini_amd.BfsProblem(g, srcs[0])
        for rep in range(2):
            for s in srcs:
                st = bfs.run(s)
                want = oracle.bfs_cpu(ro, ci, s)
                assert np.array_equal(bfs.labels(), want), (plan, rep, s)
                assert st["m_t"] == int(d[want >= 0].sum()) and (d[s] == 0 or st["levels"] == int(want.max()) + 1), (plan, rep, s, st)
        for rot in range(len(srcs)):
            batch = srcs[rot:] + srcs[:rot]
            sts, reruns = bfs.run_many(batch, mini_amd.MGX_BFS_PUSH, 0.0)
            assert np.array_equal(bfs.labels(), oracle.bfs_cpu(ro, ci, batch[-1])), (plan, rot)
            results[(plan, rot)] = [(x["m_t"], x["reached"], x["levels"]) for x in sts]
            if rot >= 2:
                assert reruns == 0, (plan, rot, reruns)
    for rot in range(len(srcs)):
        assert results[("1", rot)] == results[("0", rot)], rot


@pytest.mark.parametrize("layout", [False, True])
def test_bfs_mid_size_levels_take_m_launches(gpu_ctx, oracle, layout, monkeypatch):
    """M launches (bfs_fused_mini.hpp: a mid-size level as one launch of 64 workgroups, claims by atomicOr, no sweep):
    graphs whose levels sit between the one-workgroup chain (4096 edges, 1536 early) and the mid-size limits (131072 edges,
    32768 early) -- hubs of a few thousand edges (long rows staged in LDS, 64-edge units), tens of thousands of short rows,
    more winners than a workgroup's list holds (flushes) -- and levels ABOVE the limits behind an M launch (forwarded to
    the next slot); labels and counters against the oracle, with the launches switched off as the cross-check"""
    import mini_amd
    monkeypatch.setenv("MGX_BFS_MINI", "2")         # (by default only graphs of 2^22 vertices and more get M launches)
    rng = np.random.default_rng(2024)
    used = 0
    for trial in range(6):
        if trial < 2:       # a few hubs of 2 000 .. 6 000 edges each, leaves with a couple of edges among themselves
            h, n = 6 + trial * 10, 60000
            deg = rng.integers(2000, 6000, size=h)
            t0 = np.concatenate([np.repeat(np.arange(h), deg), rng.integers(h, n, size=30000)]).astype(np.int32)
            t1 = np.concatenate([rng.integers(h, n, size=int(deg.sum())), rng.integers(h, n, size=30000)]).astype(np.int32)
        elif trial < 4:     # sparse uniform random graphs: many levels of a few thousand short rows each
            n = 40000 + 20000 * trial
            e = int(1.6 * n)
            t0 = rng.integers(0, n, size=e).astype(np.int32); t1 = rng.integers(0, n, size=e).astype(np.int32)
        else:               # R-MAT 15 / 16: a mid-size second level in front of the big ones, mid-size stragglers behind
            n, ro, ci, w = oracle.rmat_csr(11 + trial, 8, 500 + trial)
        if trial < 4:
            ro, ci, w = oracle.csr_from_tuples(n, t0, t1, None, undir=True)
        g = _graph(gpu_ctx, ro, ci)
        if layout:
            g.build_layout()
        deg = np.diff(ro)
        bfs = mini_amd.BfsProblem(g, 0)
        srcs = [int(np.argmax(deg))] + [int(v) for v in rng.choice(np.where(deg > 0)[0], size=4, replace=False)]
        for src in srcs:
            want = oracle.bfs_cpu(ro, ci, src)
            for rep in range(2):            # (the second run has the first one's level structure as its hint)
                st = bfs.run(src)
                assert np.array_equal(bfs.labels(), want), (trial, src, rep)
                assert st["m_t"] == int(deg[want >= 0].sum()) and st["reached"] == int((want >= 0).sum()), (trial, src, rep, st)
                used += st["mini_slots"]
        sts, _ = bfs.run_many(srcs, mini_amd.MGX_BFS_PUSH, 0.0)
        assert np.array_equal(bfs.labels(), oracle.bfs_cpu(ro, ci, srcs[-1]))
        for s, st in zip(srcs, sts):
            want = oracle.bfs_cpu(ro, ci, s)
            assert st["reached"] == int((want >= 0).sum()) and st["levels"] == int(want.max()) + 1
    assert used > 0


def _check_shortest_path_tree(ro, ci, w, dist, pred, src):
    """pred is a shortest-path tree of dist: pred[src] = pred[unreached] = -1; every other reached v has an edge pred[v] -> v
    with dist[pred[v]] + w == dist[v] (float32, as the loop adds); following preds from any reached vertex ends at src"""
    n = len(ro) - 1
    inf = np.float32(3.402823466e+38)
    reached = dist < inf
    assert pred[src] == -1 and dist[src] == 0
    assert np.all(pred[~reached] == -1)
    rest = reached.copy(); rest[src] = False
    assert np.all(pred[rest] >= 0), "a reached vertex without a predecessor"
    # tight edge pred[v] -> v: among the entries of row pred[v] that point to v, one with the right weight
    srcs = np.repeat(np.arange(n, dtype=np.int64), np.diff(ro))
    key = srcs * n + ci.astype(np.int64)
    tight = (dist[srcs] + w.astype(np.float32)).astype(np.float32) == dist[ci]
    tight &= reached[srcs]
    tkeys = np.unique(key[tight])
    vs = np.nonzero(rest)[0]
    want = pred[vs].astype(np.int64) * n + vs
    assert np.all(np.isin(want, tkeys)), "a predecessor edge that is not tight (or not an edge)"
    # acyclic: pointer doubling -- after ceil(log2 n) + 1 rounds everybody reached stands at the source
    p = pred.astype(np.int64).copy()
    p[src] = src
    p[~reached] = np.arange(n)[~reached]
    for _ in range(int(np.ceil(np.log2(max(n, 2)))) + 1):
        p = p[p]
    assert np.all(p[reached] == src), "the predecessors contain a cycle"


@pytest.mark.parametrize("layout", [False, True])
def test_sssp_fused_preds_form_a_shortest_path_tree(gpu_ctx, oracle, layout):
    """sssp_functor.hxx:31-34 keeps preds in the relaxation (racy; tests/sssp/test_sssp.cu:44-51 compares them exactly); the fused
    loop builds them from its distances (mgx/sssp_preds.hpp): a tree of tight edges, the same on every call.  R-MAT graphs with
    weights in [0, 64) (zeros: equal-distance ties), real-valued weights, a directed graph, and a graph of zero weights only"""
    import mini_amd
    rng = np.random.default_rng(11)
    cases = []
    for scale, ef, seed in ((10, 8, 3), (13, 16, 5), (15, 16, 9)):
        n, ro, ci, w = oracle.rmat_csr(scale, ef, seed)
        cases.append((ro, ci, w))
    n, ro, ci, w = oracle.rmat_csr(12, 8, 21)
    cases.append((ro, ci, (rng.random(len(ci)) * 5.0).astype(np.float32)))             # real weights (not symmetric: a directed weighting)
    cases.append((ro, ci, (rng.integers(0, 2, len(ci))).astype(np.float32)))           # half the edges weigh nothing
    cases.append((ro, ci, np.zeros(len(ci), dtype=np.float32)))                        # all ties: the rounds are a BFS
    # a directed chain with zero-weight shortcuts back (tight edges in both directions between equal-distance vertices)
    k = 300
    rows = [[] for _ in range(k)]
    for i in range(k - 1):
        rows[i].append((i + 1, 0.0 if i % 3 else 2.0))
        rows[i + 1].append((i, 0.0))
    ro_c = np.zeros(k + 1, dtype=np.int32); ci_c, w_c = [], []
    for i, r in enumerate(rows):
        r.sort()
        ro_c[i + 1] = ro_c[i] + len(r)
        ci_c += [d for d, _ in r]; w_c += [x for _, x in r]
    cases.append((ro_c, np.array(ci_c, dtype=np.int32), np.array(w_c, dtype=np.float32)))
    for ci_case, (ro, ci, w) in enumerate(cases):
        g = _graph(gpu_ctx, ro, ci, w)
        if layout:
            g.build_layout(weights=True)
        deg = np.diff(ro)
        sssp = mini_amd.SsspProblem(g, 0)
        for src in (int(np.argmax(deg)), int(np.where(deg > 0)[0][-1]), 0):
            sssp.run(src)
            dist = sssp.distances()
            assert np.array_equal(dist, oracle.sssp_dijkstra_f32(ro, ci, w, src))
            st = sssp.build_preds()
            pred = sssp.preds()
            _check_shortest_path_tree(ro, ci, w, dist, pred, src)
            if ci_case >= 4:
                assert st["ties"] > 0 and st["rounds"] >= 2, st
            sssp.run(src)
            assert np.array_equal(sssp.preds(), pred), "the predecessors differ between two runs"
        sssp.close()


@pytest.mark.parametrize("slices", [0, 20, 52])
def test_neighbour_reduce_many_slices_on_a_mid_size_graph(gpu_ctx, oracle, torch_mod, monkeypatch, slices):
    """R-MAT 21 (2 M vertices: 52 slices of 40 000 in its id range) with the library's own cut (16 hot slices + tail: the fold takes a
    row's ranges in ONE chunk), with 20 and with 52 hot slices (round 6, graphs above R-MAT 22: the fold goes round 2 / 4 times,
    NRS_FOLD_CHUNK ranges at a time; 52 leaves the tail empty).  Integer min / max exact, float sums of small integers exact, against the
    oracle's serial reduce on the same CSR."""
    import mini_amd
    from mini_amd import rmat
    torch = torch_mod
    if slices:
        monkeypatch.setenv("MGX_NR_SLICES", str(slices))
    g = rmat.rmat_csr(gpu_ctx, 21, 16, seed=2121)
    n = g["n"]
    graph = mini_amd.Graph.from_device(gpu_ctx, n, g["m"], g["row_offsets"], g["col_indices"]).build_layout()
    ro, ci = g["row_offsets"].cpu().numpy(), g["col_indices"].cpu().numpy()
    rng = np.random.default_rng(2100 + slices)
    ids = np.arange(n, dtype=np.int32)
    f = mini_amd.Frontier(gpu_ctx, n).load(ids)
    vals = rng.integers(0, 4, size=n).astype(np.float32)
    red = torch.full((n,), -1, dtype=torch.float32, device="cuda")
    nz = mini_amd.segreduce(graph, f, torch.from_numpy(vals).cuda(), 0.0, red, "f32_plus")
    want, wnz = oracle.neighbor_reduce_f32_plus(ro, ci, ids, vals, 0.0)
    assert nz == wnz == len(ci)
    assert np.array_equal(red.cpu().numpy(), want)
    for op in ("i32_min", "i32_max"):
        ivals = rng.integers(-100000, 100000, size=n).astype(np.int32)
        ident = 2**31 - 1 if op == "i32_min" else -2**31
        ired = torch.full((n,), 12345, dtype=torch.int32, device="cuda")
        nz = mini_amd.segreduce(graph, f, torch.from_numpy(ivals).cuda(), ident, ired, op)
        want, wnz = oracle.neighbor_reduce_i32(ro, ci, ids, ivals, ident, op == "i32_max")
        assert nz == wnz
        assert np.array_equal(ired.cpu().numpy(), want)
    info = graph.nr_slices_info()
    if os.environ.get("MGX_NR_SLICED", "1") != "0":
        assert info["hot_slices"] == (slices if slices else 16)
        assert (info["tail_mini_units"] > 0) == (slices != 52)


@pytest.mark.parametrize("op", ["f32_plus", "i32_min", "i32_max"])
@pytest.mark.parametrize("slices", [0, 1, 2])
def test_neighbour_reduce_sliced_long_rows(gpu_ctx, oracle, torch_mod, monkeypatch, op, slices):
    """the long rows by slice of their destinations (mgx/nreduce.hpp: k_nrs_edges / k_nrs_fold, round 5) on R-MAT 17 (131 072
    vertices: four hot slices of 40 000): the default cut, and with one / two hot slices only, so that the TAIL (32-bit ids, values
    gathered) carries most of the entries; every cut against the oracle's serial reduce -- exact for ints and for float sums of
    small integers, 2e-5 relative for real-valued floats -- and the library says that the slices were built and used"""
    import mini_amd
    torch = torch_mod
    if slices:
        monkeypatch.setenv("MGX_NR_SLICES", str(slices))
        # ... and the fold's three tiers (a workgroup / a wave / a thread per row) moved down to where this graph has rows
        monkeypatch.setenv("MGX_NR_FOLD_DEGS", "64/48/40" if slices == 1 else "2000/500/100")
    n, ro, ci, w = oracle.rmat_csr(17, 16, 77)
    g = _graph(gpu_ctx, ro, ci).build_layout()
    rng = np.random.default_rng(17 + slices)
    ids = np.arange(n, dtype=np.int32)
    f = mini_amd.Frontier(gpu_ctx, n).load(ids)
    for rep in range(2):
        if op == "f32_plus":
            for real in (False, True):
                vals = (rng.random(n) * 3.0).astype(np.float32) if real else rng.integers(0, 8, size=n).astype(np.float32)
                dv = torch.from_numpy(vals).cuda()
                red = torch.full((n,), -1, dtype=torch.float32, device="cuda")
                nz = mini_amd.segreduce(g, f, dv, 0.0, red, op)
                want, wnz = oracle.neighbor_reduce_f32_plus(ro, ci, ids, vals, 0.0)
                assert nz == wnz == len(ci)
                got = red.cpu().numpy()
                if real:
                    assert np.allclose(got, want, rtol=2e-5, atol=1e-6), np.abs(got - want).max()
                else:
                    assert np.array_equal(got, want)
        else:
            vals = rng.integers(-1000, 1000, size=n).astype(np.int32)
            dv = torch.from_numpy(vals).cuda()
            ident = 2**31 - 1 if op == "i32_min" else -2**31
            red = torch.full((n,), 12345, dtype=torch.int32, device="cuda")
            nz = mini_amd.segreduce(g, f, dv, ident, red, op)
            want, wnz = oracle.neighbor_reduce_i32(ro, ci, ids, vals, ident, op == "i32_max")
            assert nz == wnz
            assert np.array_equal(red.cpu().numpy(), want)
    info = g.nr_slices_info()
    if os.environ.get("MGX_NR_SLICED", "1") != "0":
        assert info["mini_units"] > 0 and info["long_rows"] > 0
        assert info["hot_slices"] == (slices if slices else 4)
        assert (info["tail_mini_units"] > 0) == (slices in (1, 2))
