"""GPU suite (-m gpu): the pull-direction operators one call at a time -- gen_unvisited_kernel, sparse_to_dense_kernel,
advance_backward_kernel (gunrock/src/advance.hxx:69-160) and the filter between two bottom-up steps
(bfs_enactor.hxx:74-112) -- against the oracle's serial restatements, superstep by superstep.  A vertex is claimed in a
step iff one of its in-neighbours is in the frontier bitmap, so every array is deterministic: bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _transpose(n, ro, ci):
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(ro))
    order = np.lexsort((rows, ci))
    co = np.concatenate([[0], np.cumsum(np.bincount(ci, minlength=n))]).astype(np.int32)
    return co, rows[order].astype(np.int32)


def _directed_rmat(oracle, scale, ef, seed):
    n = 1 << scale
    s, d, _ = oracle.rmat_edges(scale, 0, ef * n, seed)
    ro, ci, _ = oracle.csr_from_tuples(n, s, d, None, False)
    return n, ro, ci


@pytest.mark.parametrize("push_levels", [1, 2, 3])
@pytest.mark.parametrize("kind", ["undirected12", "directed12", "undirected15"])
def test_pull_operators_superstep_by_superstep(gpu_ctx, oracle, kind, push_levels):
    import mini_amd
    if kind.startswith("undirected"):
        n, ro, ci, _ = oracle.rmat_csr(int(kind[10:]), 16, 5)
        co, ri = ro, ci                                   # the reference's "CSC" is the CSR again
        g = mini_amd.Graph.from_host(gpu_ctx, ro, ci, None)
    else:
        n, ro, ci = _directed_rmat(oracle, int(kind[8:]), 16, 9)
        co, ri = _transpose(n, ro, ci)                    # a genuine CSC
        g = mini_amd.Graph.from_host(gpu_ctx, ro, ci, None, co, ri)
    m = len(ci)
    src = int(np.argmax(np.diff(ro)))
    bfs = mini_amd.BfsProblem(g, src)
    cap = max(m, n)
    fa, fb = mini_amd.Frontier(gpu_ctx, cap), mini_amd.Frontier(gpu_ctx, cap)
    fa.load(np.array([src], dtype=np.int32))
    labels = np.full(n, -1, dtype=np.int32)
    labels[src] = 0
    front = np.array([src], dtype=np.int32)
    it = 0
    for it in range(push_levels):                         # top-down levels, both sides in lockstep
        bfs.advance(fa, fb, it)
        raw = oracle.bfs_advance(ro, ci, labels, front, it)
        kept = bfs.filter(fb, fa, it)
        front = np.sort(oracle.bfs_filter(raw))
        assert kept == len(front) and np.array_equal(np.sort(fa.read()), front)
        fa.load(front)
    assert np.array_equal(bfs.labels(), labels) and len(front) > 0
    it = push_levels                                      # the frontier's label

    # gen_unvisited over the iota (bfs_enactor.hxx:80)
    iota = mini_amd.Frontier(gpu_ctx, n).fill_iota(n)
    unv, unv2 = mini_amd.Frontier(gpu_ctx, n), mini_amd.Frontier(gpu_ctx, n)
    o_unv = oracle.bfs_gen_unvisited(labels, np.arange(n, dtype=np.int32))
    assert bfs.gen_unvisited(iota, unv, it) == len(o_unv)
    assert np.array_equal(unv.read(), o_unv)              # stable: ascending ids
    # sparse_to_dense of the frontier into a zeroed n-int bitmap (:86-91)
    bm, bm2 = mini_amd.Frontier(gpu_ctx, cap).fill(0, n), mini_amd.Frontier(gpu_ctx, cap)
    o_bm = np.zeros(n, dtype=np.int32)
    bfs.sparse_to_dense(fa, bm, it)
    oracle.bfs_sparse_to_dense(labels, front, o_bm, it)
    assert np.array_equal(bm.read(), o_bm) and o_bm.sum() == len(front)

    steps = 0
    while len(o_unv):
        bm2.fill(0, n)
        o_bm2 = np.zeros(n, dtype=np.int32)
        inspected = bfs.advance_backward(unv, bm, bm2, it)
        o_unv = o_unv.copy()
        o_inspected = oracle.bfs_advance_backward(co, ri, labels, o_unv, o_bm, o_bm2, it)
        assert inspected == o_inspected
        assert np.array_equal(bfs.labels(), labels)
        assert np.array_equal(bm2.read(), o_bm2)
        assert np.array_equal(unv.read(), o_unv)          # claimed slots are -1 on both sides
        kept = bfs.filter(unv, unv2, it)                  # cond_filter: idx != -1 (bfs_functor.hxx)
        o_next = oracle.bfs_filter(o_unv)
        assert kept == len(o_next) and np.array_equal(unv2.read(), o_next)
        steps += 1
        if len(o_next) == len(o_unv):                     # nobody claimed: the rest is unreachable
            break
        o_unv, o_bm = o_next, o_bm2
        unv, unv2 = unv2, unv
        bm, bm2 = bm2, bm
        it += 1
    assert steps >= 1
    # together the push levels and the bottom-up steps are a whole BFS (in-edges == transposed out-edges)
    assert np.array_equal(labels, oracle.bfs_cpu(ro, ci, src))
