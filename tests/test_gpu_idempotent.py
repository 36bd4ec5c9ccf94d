"""The reference's IDEMPOTENT traversal mode (advance.hxx:60 with idempotence = true, filter.hxx:33-119)
through the C-ABI: mgx_bfs_advance_idempotent, mgx_bfs_uniquify, mgx_bfs_enact_idempotent.  Upstream has no enactor and
no test for it; parity is by labels against the oracle's bfs_problem_t::cpu."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _neighbours(ro, ci, frontier):
    return np.concatenate([ci[ro[v]:ro[v + 1]] for v in frontier]) if len(frontier) else np.zeros(0, dtype=np.int32)


@pytest.mark.parametrize("scale", [8, 10, 13, 16])
def test_idempotent_supersteps_against_the_oracle(gpu_ctx, oracle, scale):
    """superstep by superstep: advance<idempotence> emits EVERY neighbour of the frontier in (frontier, row) order;
    uniquify keeps exactly one copy of each vertex at depth level + 1, in input order, and labels it"""
    import mini_amd
    n, ro, ci, w = oracle.rmat_csr(scale, 16, 100 + scale)
    g = mini_amd.Graph.from_host(gpu_ctx, ro, ci)
    deg = np.diff(ro)
    for src in (int(np.argmax(deg)), 0 if deg[0] else int(np.where(deg > 0)[0][0]), int(np.where(deg > 0)[0][-1])):
        want = oracle.bfs_cpu(ro, ci, src)
        bfs = mini_amd.BfsProblem(g, src)
        bfs.reset(src)
        fa, fb = mini_amd.Frontier(gpu_ctx, len(ci) + 1), mini_amd.Frontier(gpu_ctx, len(ci) + 1)
        fa.load(np.array([src], dtype=np.int32))
        frontier = np.array([src], dtype=np.int32)
        for it in range(int(want.max()) + 2):
            front = bfs.advance_idempotent(fa, fb, it)
            raw = fb.read()
            exp = _neighbours(ro, ci, frontier)
            assert front == len(exp) and np.array_equal(raw, exp), (src, it)
            if front == 0:
                break
            kept = bfs.uniquify(fb, fa, it)
            out = fa.read()
            level = np.where(want == it + 1)[0]
            assert kept == len(level) and np.array_equal(np.sort(out), level), (src, it)
            # survivors keep input order: `out` is a subsequence of `raw`
            pos, j = [], 0
            for i, v in enumerate(raw.tolist()):
                if j < len(out) and v == out[j]:
                    pos.append(i); j += 1
            assert j == len(out), (src, it)
            frontier = out
            if kept == 0:
                break
        assert np.array_equal(bfs.labels(), want), src


@pytest.mark.parametrize("name,undir", [("bfs_test.mtx", True), ("sssp_test.mtx", False), ("kcore_test.mtx", True),
                                        ("synthetic_dup.mtx", False), ("pr_test.mtx", True)])
def test_enact_idempotent_on_the_reference_fixtures(gpu_ctx, oracle, name, undir):
    import mini_amd
    n, ro, ci, w = mini_amd.load_mtx(os.path.join(GOLD, name), undir=undir)
    g = mini_amd.Graph.from_host(gpu_ctx, ro, ci, w)
    for src in range(n):
        bfs = mini_amd.BfsProblem(g, src)
        st = bfs.enact_idempotent()
        want = oracle.bfs_cpu(ro, ci, src)
        assert np.array_equal(bfs.labels(), want), (name, src)
        assert st["edges"] == int(np.diff(ro)[want >= 0].sum())


@pytest.mark.parametrize("scale", [8, 12, 16])
def test_enact_idempotent_rmat_and_repeated_runs(gpu_ctx, oracle, scale):
    """labels equal the oracle's and the CAS path's (enact_pushpull); the handle is reusable: reset + run again"""
    import mini_amd
    n, ro, ci, w = oracle.rmat_csr(scale, 16, 7 * scale)
    g = mini_amd.Graph.from_host(gpu_ctx, ro, ci)
    deg = np.diff(ro)
    bfs = mini_amd.BfsProblem(g, 0)
    iso = np.where(deg == 0)[0]
    srcs = [int(np.argmax(deg)), 0, int(np.where(deg > 0)[0][7])] + ([int(iso[0])] if len(iso) else [])
    for rnd in range(2):
        for src in srcs:
            want = oracle.bfs_cpu(ro, ci, src)
            bfs.reset(src)
            st = bfs.enact_idempotent()
            assert np.array_equal(bfs.labels(), want), (src, rnd)
            assert st["edges"] == int(deg[want >= 0].sum())            # every edge of a reached vertex expanded once
            bfs.reset(src)
            bfs.enact_pushpull()
            assert np.array_equal(bfs.labels(), want)


def test_uniquify_duplicates_holes_and_vertex_zero(gpu_ctx, oracle):
    """one call on a hand-made frontier: -1 holes dropped, duplicates (inside one wave, across waves, across tiles)
    collapse to one copy, the source never comes back, vertex 0 is an ordinary vertex (upstream's culls skip it)"""
    import mini_amd
    n = 5000
    t0 = np.arange(0, n - 1, dtype=np.int32); t1 = np.arange(1, n, dtype=np.int32)
    ro, ci, w = oracle.csr_from_tuples(n, t0, t1, None, undir=True)
    g = mini_amd.Graph.from_host(gpu_ctx, ro, ci)
    src = 17
    bfs = mini_amd.BfsProblem(g, src)
    bfs.reset(src)
    rng = np.random.default_rng(3)
    items = np.concatenate([np.full(300, 0), np.full(200, src), np.arange(100, 164).repeat(3), rng.integers(0, n, 9000),
                            np.full(50, -1), np.array([4999, 4999, 0, 1, 1])]).astype(np.int32)
    rng.shuffle(items)
    fin, fout = mini_amd.Frontier(gpu_ctx, len(items)), mini_amd.Frontier(gpu_ctx, len(items))
    fin.load(items)
    kept = bfs.uniquify(fin, fout, 4)
    out = fout.read()
    want = np.setdiff1d(np.unique(items[items >= 0]), [src])
    assert kept == len(want) and np.array_equal(np.sort(out), want)
    lab = bfs.labels()
    assert np.all(lab[want] == 5) and lab[src] == 0 and np.all(np.delete(lab, np.append(want, src)) == -1)
    # a second call sees nothing new
    assert bfs.uniquify(fin, fout, 5) == 0
    # after a reset only the source is marked
    bfs.reset(src)
    assert bfs.uniquify(fin, fout, 0) == len(want)
