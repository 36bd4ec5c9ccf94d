"""Host-side mirror of the reference's data model and operator interface over the C-ABI.

Names follow the reference (gunrock/src/*.hxx): Graph ~ graph_device_t, Frontier ~ frontier_t<int>,
BfsProblem ~ bfs_problem_t + bfs_enactor_t, SsspProblem ~ sssp_problem_t + sssp_enactor_t,
PrProblem ~ pr_problem_t + pr_enactor_t, KcoreProblem ~ kcore_problem_t + kcore_enactor_t.  Every method is one C-ABI call; nothing is computed here.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, lib


def _np_i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _dev_ptr(x):
    """int device address, or an object with data_ptr() (torch tensor), or None"""
    if x is None:
        return None
    if hasattr(x, "data_ptr"):
        return C.c_void_p(x.data_ptr())
    return C.c_void_p(int(x))


class Context:
    """standard_context_t: one device, one stream, scratch arena (tests/bfs/test_bfs.cu:22)."""

    def __init__(self, device=0, stream=None):
        h = C.c_void_p()
        check(lib.mgx_ctx_create(int(device), C.c_void_p(stream or 0), C.byref(h)))
        self._h = h
        self.device = device

    def set_stream(self, stream):
        check(lib.mgx_ctx_set_stream(self._h, C.c_void_p(stream or 0)))

    def synchronize(self):
        check(lib.mgx_ctx_synchronize(self._h))

    @property
    def num_cus(self):
        v = C.c_int()
        check(lib.mgx_ctx_num_cus(self._h, C.byref(v)))
        return v.value

    def close(self):
        if self._h:
            lib.mgx_ctx_destroy(self._h)
            self._h = None


class Graph:
    """graph_device_t (graph.hxx:37-83).  CSC slots mirror the CSR unless a CSC is given."""

    def __init__(self, ctx, handle, keepalive=None):
        self.ctx, self._h, self._keep = ctx, handle, keepalive
        n, m = C.c_int(), C.c_int64()
        check(lib.mgx_graph_dims(self._h, C.byref(n), C.byref(m)))
        self.num_nodes, self.num_edges = n.value, m.value

    @classmethod
    def from_host(cls, ctx, row_offsets, col_indices, weights=None, col_offsets=None, row_indices=None,
                  row_weights=None):
        ro, ci = _np_i32(row_offsets), _np_i32(col_indices)
        w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float32)
        co = None if col_offsets is None else _np_i32(col_offsets)
        ri = None if row_indices is None else _np_i32(row_indices)
        rw = None if row_weights is None else np.ascontiguousarray(row_weights, dtype=np.float32)
        h = C.c_void_p()
        check(lib.mgx_graph_upload(ctx._h, len(ro) - 1, len(ci), _ptr(ro), _ptr(ci), _ptr(w), _ptr(co), _ptr(ri),
                                   _ptr(rw), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def from_device(cls, ctx, num_nodes, num_edges, row_offsets, col_indices, weights=None, col_offsets=None,
                    row_indices=None, row_weights=None):
        """Borrow device arrays (torch tensors or raw addresses); they are kept alive by the Graph."""
        h = C.c_void_p()
        check(lib.mgx_graph_wrap_device(ctx._h, int(num_nodes), int(num_edges), _dev_ptr(row_offsets),
                                        _dev_ptr(col_indices), _dev_ptr(weights), _dev_ptr(col_offsets),
                                        _dev_ptr(row_indices), _dev_ptr(row_weights), C.byref(h)))
        return cls(ctx, h, keepalive=(row_offsets, col_indices, weights, col_offsets, row_indices, row_weights))

    def attach_layout(self, layout_row_offsets, layout_col_indices, new_of_old, old_of_new, layout_weights=None):
        """Hub-first (degree-descending) relabelled CSR + id maps for the fused traversals (device tensors);
        layout_weights (optional, the layout's edge order) lets the fused SSSP loop run in layout space too."""
        check(lib.mgx_graph_attach_layout(self._h, _dev_ptr(layout_row_offsets), _dev_ptr(layout_col_indices),
                                          _dev_ptr(new_of_old), _dev_ptr(old_of_new)))
        if layout_weights is not None:
            check(lib.mgx_graph_attach_layout_weights(self._h, _dev_ptr(layout_weights)))
        self._layout = (layout_row_offsets, layout_col_indices, new_of_old, old_of_new, layout_weights)
        return self

    def build_layout(self, weights=False):
        """The same layout, built by the library from the graph's own CSR (mgx_graph_build_layout) and owned by it."""
        check(lib.mgx_graph_build_layout(self._h, int(bool(weights))))
        return self

    def layout_info(self):
        """what the layout holds (mgx_graph_layout_info): unit blocks, cold-edge lists, device bytes"""
        out = (C.c_int64 * 8)()
        check(lib.mgx_graph_layout_info(self._h, out))
        keys = ("has_layout", "units", "units_24bit", "cold_pairs", "cold_slices", "hot_units", "cold_majority", "device_bytes")
        return dict(zip(keys, (int(x) for x in out)))

    def nr_slices_info(self):
        """the long rows by slice of their destinations (mgx_graph_nr_slices_info; built at the first full-frontier reduce)"""
        out = (C.c_int64 * 5)()
        check(lib.mgx_graph_nr_slices_info(self._h, out))
        keys = ("mini_units", "hot_slices", "long_rows", "multi_lane_fold_rows", "tail_mini_units")
        return dict(zip(keys, (int(x) for x in out)))

    def build_csc(self):
        """Genuine CSC (transpose) built by the library on the device (mgx_graph_build_csc): in-edges for the bottom-up
        levels on directed graphs."""
        check(lib.mgx_graph_build_csc(self._h))
        return self

    def csc_arrays(self):
        """(col_offsets, row_indices, row_values) of the CSC slots as host numpy arrays"""
        n, m = self.num_nodes, self.num_edges
        co, ri = np.empty(n + 1, dtype=np.int32), np.empty(max(m, 1), dtype=np.int32)
        rv = np.empty(max(m, 1), dtype=np.float32)
        check(lib.mgx_graph_csc_read(self._h, co.ctypes.data_as(C.c_void_p), ri.ctypes.data_as(C.c_void_p),
                                     rv.ctypes.data_as(C.c_void_p)))
        return co, ri[:m], rv[:m]

    def layout_arrays(self, weights=False):
        """(layout_row_offsets, layout_col_indices, new_of_old, old_of_new[, layout_weights]) as host numpy arrays"""
        n, m = self.num_nodes, self.num_edges
        lro, lci = np.empty(n + 1, dtype=np.int32), np.empty(max(m, 1), dtype=np.int32)
        n2o, o2n = np.empty(n, dtype=np.int32), np.empty(n, dtype=np.int32)
        lw = np.empty(max(m, 1), dtype=np.float32) if weights else None
        check(lib.mgx_graph_layout_read(self._h, lro.ctypes.data_as(C.c_void_p), lci.ctypes.data_as(C.c_void_p),
                                        n2o.ctypes.data_as(C.c_void_p), o2n.ctypes.data_as(C.c_void_p),
                                        lw.ctypes.data_as(C.c_void_p) if weights else None))
        out = (lro, lci[:m], n2o, o2n)
        return out + (lw[:m],) if weights else out

    def close(self):
        if self._h:
            lib.mgx_graph_free(self._h)
            self._h = None


def load_mtx(path, undir=False, random_edge_value=False, genuine_csc=None):
    """load_graph (graph.hxx:96-223) -> (n, row_offsets, col_indices, weights) on the host; with genuine_csc given
    (True / False) also the loader's CSC slots: (..., col_offsets, row_indices, row_weights)."""
    n, m = C.c_int(), C.c_int64()
    ptrs = [C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_float)()]
    if genuine_csc is None:
        check(lib.mgx_load_mtx(str(path).encode(), int(undir), int(random_edge_value), C.byref(n), C.byref(m),
                               *[C.byref(p) for p in ptrs]))
    else:
        ptrs += [C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_float)()]
        check(lib.mgx_load_mtx_csc(str(path).encode(), int(undir), int(random_edge_value), int(bool(genuine_csc)),
                                   C.byref(n), C.byref(m), *[C.byref(p) for p in ptrs]))
    N, M = n.value, m.value
    try:
        out = []
        for i, p in enumerate(ptrs):
            cnt = N + 1 if i % 3 == 0 else M
            out.append(np.ctypeslib.as_array(p, (max(cnt, 1),))[:cnt].copy())
    finally:
        for p in ptrs:
            lib.mgx_host_free(p)
    return (N,) + tuple(out)


def save_csr_cache(path, row_offsets, col_indices, weights=None, csc=None, undirected=False):
    """binary CSR cache (mgx_graph_save_csr); csc = (col_offsets, row_indices, row_weights) to store a genuine CSC too"""
    ro, ci = _np_i32(row_offsets), _np_i32(col_indices)
    w = None if weights is None else np.ascontiguousarray(weights, dtype=np.float32)
    co = ri = rw = None
    if csc is not None:
        co, ri = _np_i32(csc[0]), _np_i32(csc[1])
        rw = None if csc[2] is None else np.ascontiguousarray(csc[2], dtype=np.float32)
    check(lib.mgx_graph_save_csr(str(path).encode(), len(ro) - 1, len(ci), int(bool(undirected)), _ptr(ro), _ptr(ci), _ptr(w),
                                 _ptr(co), _ptr(ri), _ptr(rw)))


def load_csr_cache(path):
    """-> dict(n, undirected, row_offsets, col_indices, weights, csc = (col_offsets, row_indices, row_weights) | None)"""
    n, m, u = C.c_int(), C.c_int64(), C.c_int()
    ptrs = [C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_float)(), C.POINTER(C.c_int)(), C.POINTER(C.c_int)(), C.POINTER(C.c_float)()]
    check(lib.mgx_graph_load_csr(str(path).encode(), C.byref(n), C.byref(m), C.byref(u), *[C.byref(p) for p in ptrs]))
    N, M = n.value, m.value
    try:
        arrs = []
        for i, p in enumerate(ptrs):
            cnt = N + 1 if i % 3 == 0 else M
            arrs.append(np.ctypeslib.as_array(p, (max(cnt, 1),))[:cnt].copy() if p else None)
    finally:
        for p in ptrs:
            if p:
                lib.mgx_host_free(p)
    return {"n": N, "undirected": bool(u.value), "row_offsets": arrs[0], "col_indices": arrs[1], "weights": arrs[2],
            "csc": None if arrs[3] is None else (arrs[3], arrs[4], arrs[5])}


class Frontier:
    """frontier_t<int> (frontier.hxx:12-99)."""

    def __init__(self, ctx, capacity):
        h = C.c_void_p()
        check(lib.mgx_frontier_create(ctx._h, int(capacity), C.byref(h)))
        self.ctx, self._h = ctx, h

    def load(self, ids):
        a = _np_i32(ids)
        check(lib.mgx_frontier_load(self._h, _ptr(a), len(a)))
        return self

    def fill_iota(self, n):
        check(lib.mgx_frontier_fill_iota(self._h, int(n)))
        return self

    def fill(self, value, n):
        check(lib.mgx_frontier_fill(self._h, int(value), int(n)))
        return self

    def resize(self, n):
        check(lib.mgx_frontier_resize(self._h, int(n)))

    @property
    def size(self):
        v = C.c_int64()
        check(lib.mgx_frontier_size(self._h, C.byref(v)))
        return v.value

    @property
    def capacity(self):
        v = C.c_int64()
        check(lib.mgx_frontier_capacity(self._h, C.byref(v)))
        return v.value

    @property
    def device_ptr(self):
        p = C.c_void_p()
        check(lib.mgx_frontier_device_ptr(self._h, C.byref(p)))
        return p.value

    def read(self):
        n = self.size
        out = np.empty(max(n, 1), dtype=np.int32)
        got = C.c_int64()
        check(lib.mgx_frontier_read(self._h, _ptr(out), len(out), C.byref(got)))
        return out[:got.value]

    def close(self):
        if self._h:
            lib.mgx_frontier_free(self._h)
            self._h = None


def _i64():
    return C.c_int64()


class BfsProblem:
    """bfs_problem_t + bfs_enactor_t (gunrock/src/bfs/)."""

    def __init__(self, graph, src=0):
        h = C.c_void_p()
        check(lib.mgx_bfs_create(graph._h, int(src), C.byref(h)))
        self.graph, self._h = graph, h

    def reset(self, src):
        check(lib.mgx_bfs_reset(self._h, int(src)))

    def labels(self):
        out = np.empty(self.graph.num_nodes, dtype=np.int32)
        check(lib.mgx_bfs_labels(self._h, _ptr(out)))
        return out

    def preds(self):
        out = np.empty(self.graph.num_nodes, dtype=np.int32)
        check(lib.mgx_bfs_preds(self._h, _ptr(out)))
        return out

    @property
    def labels_device_ptr(self):
        p = C.c_void_p()
        check(lib.mgx_bfs_labels_device(self._h, C.byref(p)))
        return p.value

    # operators ------------------------------------------------------------------------------
    def advance(self, fin, fout, iteration):
        v = _i64()
        check(lib.mgx_bfs_advance(self._h, fin._h, fout._h, int(iteration), C.byref(v)))
        return v.value

    def filter(self, fin, fout, iteration):
        v = _i64()
        check(lib.mgx_bfs_filter(self._h, fin._h, fout._h, int(iteration), C.byref(v)))
        return v.value

    def advance_filter_fused(self, fin, fout, iteration):
        v = _i64()
        check(lib.mgx_bfs_advance_filter_fused(self._h, fin._h, fout._h, int(iteration), C.byref(v)))
        return v.value

    def gen_unvisited(self, indices, unvisited, iteration=0):
        v = _i64()
        check(lib.mgx_bfs_gen_unvisited(self._h, indices._h, unvisited._h, int(iteration), C.byref(v)))
        return v.value

    def sparse_to_dense(self, sparse, dense, iteration):
        check(lib.mgx_bfs_sparse_to_dense(self._h, sparse._h, dense._h, int(iteration)))

    def advance_backward(self, unvisited, bitmap, bitmap_out, iteration):
        v = _i64()
        check(lib.mgx_bfs_advance_backward(self._h, unvisited._h, bitmap._h, bitmap_out._h, int(iteration),
                                           C.byref(v)))
        return v.value

    # enactors -------------------------------------------------------------------------------
    def enact_pushpull(self, threshold=None):
        """bfs_enactor_t::enact_pushpull; default threshold 1/n like test_bfs.cu:30."""
        if threshold is None:
            threshold = 1.0 / max(self.graph.num_nodes, 1)
        st = (C.c_int64 * 4)()
        check(lib.mgx_bfs_enact_pushpull(self._h, C.c_float(threshold), st))
        return {"pushed_iterations": st[0], "total_iterations": st[1], "pushed_edges": st[2], "pulled_edges": st[3]}

    def enact_idempotent(self):
        """the reference's idempotent mode as a loop: advance<idempotence> (every neighbour, no atomics) + uniquify"""
        st = (C.c_int64 * 2)()
        check(lib.mgx_bfs_enact_idempotent(self._h, st))
        return {"iterations": st[0], "edges": st[1]}

    def advance_idempotent(self, fin, fout, iteration):
        v = C.c_int64()
        check(lib.mgx_bfs_advance_idempotent(self._h, fin._h, fout._h, int(iteration), C.byref(v)))
        return v.value

    def uniquify(self, fin, fout, iteration):
        v = C.c_int64()
        check(lib.mgx_bfs_uniquify(self._h, fin._h, fout._h, int(iteration), C.byref(v)))
        return v.value

    STATS_LEN = 24

    @staticmethod
    def new_stats():
        """a buffer for run_into()"""
        return (C.c_int64 * BfsProblem.STATS_LEN)()

    def run_into(self, src, mode, alpha, st):
        """Fused device-resident traversal, counters into a caller-made buffer (new_stats()): the call a timing loop
        makes -- nothing is allocated or converted between two traversals; stats_dict(st) reads the buffer afterwards."""
        rc = lib.mgx_bfs_run_stats(self._h, src, mode, alpha, st, BfsProblem.STATS_LEN)
        if rc:
            check(rc)

    def run_many(self, sources, mode=_lib.MGX_BFS_PUSH, alpha=0.0, prepared=None):
        """A batch of sources enqueued back to back, one host wait (mgx_bfs_run_many).  Returns (list of stats dicts,
        reruns); labels() are the last source's.  prepared = prepare_many(sources): buffers made beforehand, for timing loops
        (then only the ctypes call happens here and the raw buffer is returned instead of dicts)."""
        if prepared is None:
            srcs, st, rr, count = self.prepare_many(sources)
        else:
            srcs, st, rr, count = prepared
        rc = lib.mgx_bfs_run_many(self._h, srcs, count, int(mode), float(alpha), st, BfsProblem.STATS_LEN, C.byref(rr))
        if rc:
            check(rc)
        if prepared is not None:
            return st, rr.value
        L = BfsProblem.STATS_LEN
        return [self.stats_dict(st[i * L:(i + 1) * L]) for i in range(count)], rr.value

    @staticmethod
    def prepare_many(sources):
        n = len(sources)
        return (C.c_int * max(n, 1))(*[int(s) for s in sources]), (C.c_int64 * (max(n, 1) * BfsProblem.STATS_LEN))(), C.c_int(0), n

    @staticmethod
    def stats_dict(st):
        return {"levels": st[0], "reached": st[1], "m_t": st[2], "push_edges": st[3], "pull_edges": st[4],
                "push_levels": st[5], "kernel_launches": st[6], "kernel_ns": st[7], "frontier_vertices": st[8],
                "claims": st[9], "dom_launches": st[10], "dom_ns": st[11], "dom_edges": st[12],
                "dom_vertices": st[13],
                "dom_kernel": "k_bfs_push (long rows / merged launch)" if st[14] else "k_bfs_push (short rows)",
                "small_levels": st[15], "slots": st[16], "dense_slots": st[17], "vshort_slots": st[18], "lazy_slots": st[19],
                "cold_slots": st[20], "mini_slots": st[21]}

    def run(self, src, mode=_lib.MGX_BFS_PUSH, alpha=0.0):
        """Fused device-resident traversal."""
        st = self.new_stats()
        self.run_into(int(src), int(mode), float(alpha), st)
        return self.stats_dict(st)

    def level_trace(self, cap=4096):
        nf, ne, lv = (C.c_int64 * cap)(), (C.c_int64 * cap)(), C.c_int()
        check(lib.mgx_bfs_level_trace(self._h, cap, nf, ne, C.byref(lv)))
        L = min(lv.value, cap)
        return [(nf[i], ne[i]) for i in range(L)]

    def set_kernel_timing(self, on=True):
        """events around every push-kernel launch (costs ~6 us of stream gap per event: profiling runs only).
        True / 1: the parts of a slot's push as separate launches; 2: the one merged launch a traversal really runs"""
        check(lib.mgx_bfs_set_kernel_timing(self._h, int(on)))

    def kernel_times(self):
        """per-launch timing of the two push kernels of the last run()"""
        c = (C.c_int64 * 8)()
        check(lib.mgx_bfs_kernel_times(self._h, c))
        keys = ("launches", "ns", "edges", "vertices")
        return {"stream": dict(zip(keys, c[0:4])), "wave": dict(zip(keys, c[4:8]))}

    def level_times_ms(self, cap=63):
        """per-level duration from device-side timestamps (no host synchronisation per level)"""
        ms, lv = (C.c_float * cap)(), C.c_int()
        check(lib.mgx_bfs_level_times(self._h, cap, ms, C.byref(lv)))
        return [ms[i] for i in range(min(lv.value, cap))]

    def level_kernel_times_ms(self, cap=64):
        a, b = (C.c_float * cap)(), (C.c_float * cap)()
        check(lib.mgx_bfs_level_kernel_times(self._h, cap, a, b))
        return [(a[i], b[i]) for i in range(cap)]

    def level_claims(self, cap=64):
        c = (C.c_int64 * cap)()
        check(lib.mgx_bfs_level_claims(self._h, cap, c))
        return [c[i] for i in range(cap)]

    def batch_times_ms(self, cap=256):
        ms, nb = (C.c_float * cap)(), C.c_int()
        check(lib.mgx_bfs_batch_times(self._h, cap, ms, C.byref(nb)))
        return [ms[i] for i in range(min(nb.value, cap))]

    def close(self):
        if self._h:
            lib.mgx_bfs_free(self._h)
            self._h = None


class SsspProblem:
    """sssp_problem_t + sssp_enactor_t (gunrock/src/sssp/)."""

    def __init__(self, graph, src=0):
        h = C.c_void_p()
        check(lib.mgx_sssp_create(graph._h, int(src), C.byref(h)))
        self.graph, self._h = graph, h

    def reset(self, src):
        check(lib.mgx_sssp_reset(self._h, int(src)))

    def distances(self):
        out = np.empty(self.graph.num_nodes, dtype=np.float32)
        check(lib.mgx_sssp_distances(self._h, _ptr(out)))
        return out

    def preds(self):
        """after enact(): the functor's preds; after run(): a shortest-path tree built from the distances (mgx/sssp_preds.hpp)"""
        out = np.empty(self.graph.num_nodes, dtype=np.int32)
        check(lib.mgx_sssp_preds(self._h, _ptr(out)))
        return out

    def build_preds(self):
        """builds the predecessors of the last run()'s distances on the device (no copy) -> {ties, rounds}"""
        st = (C.c_int64 * 2)()
        check(lib.mgx_sssp_build_preds(self._h, st))
        return {"ties": st[0], "rounds": st[1]}

    @property
    def distances_device_ptr(self):
        p = C.c_void_p()
        check(lib.mgx_sssp_distances_device(self._h, C.byref(p)))
        return p.value

    def advance(self, fin, fout, iteration):
        v = _i64()
        check(lib.mgx_sssp_advance(self._h, fin._h, fout._h, int(iteration), C.byref(v)))
        return v.value

    def filter(self, fin, fout, iteration):
        v = _i64()
        check(lib.mgx_sssp_filter(self._h, fin._h, fout._h, int(iteration), C.byref(v)))
        return v.value

    def enact(self, queue_sizing=1.0):
        st = (C.c_int64 * 3)()
        check(lib.mgx_sssp_enact(self._h, C.c_float(queue_sizing), st))
        return {"iterations": st[0], "relaxations": st[1], "frontier_total": st[2]}

    def set_kernel_timing(self, on=True):
        check(lib.mgx_sssp_set_kernel_timing(self._h, int(bool(on))))

    def iteration_trace(self, cap=63):
        """[(frontier vertices, edges relaxed, ms)] of the last run()'s iterations (device-side timestamps)"""
        nf, ne, ms, it = (C.c_int64 * cap)(), (C.c_int64 * cap)(), (C.c_float * cap)(), C.c_int()
        check(lib.mgx_sssp_iteration_trace(self._h, cap, nf, ne, ms, C.byref(it)))
        return [(nf[i], ne[i], ms[i]) for i in range(min(it.value, cap))]

    def kernel_times(self):
        """{launches, ns} of k_sssp_relax in the last run() (after set_kernel_timing)"""
        c = (C.c_int64 * 2)()
        check(lib.mgx_sssp_kernel_times(self._h, c))
        return {"launches": c[0], "ns": c[1]}

    def run(self, src, delta=None):
        """fused device-resident loop; delta: near / far bucket width (None / 0: plain frontier Bellman-Ford)"""
        st = (C.c_int64 * 3)()
        if delta is None:
            check(lib.mgx_sssp_run(self._h, int(src), st))
        else:
            check(lib.mgx_sssp_run_delta(self._h, int(src), C.c_float(delta), st))
        return {"iterations": st[0], "relaxations": st[1], "frontier_total": st[2]}

    def close(self):
        if self._h:
            lib.mgx_sssp_free(self._h)
            self._h = None


class PrProblem:
    """pr_problem_t + pr_enactor_t (gunrock/src/pr/)."""

    def __init__(self, graph, max_iter=10):
        h = C.c_void_p()
        check(lib.mgx_pr_create(graph._h, int(max_iter), C.byref(h)))
        self.graph, self._h, self.max_iter = graph, h, max_iter

    def enact(self):
        lens = (C.c_int64 * max(self.max_iter, 1))()
        it = C.c_int()
        check(lib.mgx_pr_enact(self._h, lens, C.byref(it)))
        return [lens[i] for i in range(it.value)]

    def ranks(self):
        out = np.empty(self.graph.num_nodes, dtype=np.float32)
        check(lib.mgx_pr_ranks(self._h, _ptr(out)))
        return out

    def close(self):
        if self._h:
            lib.mgx_pr_free(self._h)
            self._h = None


class KcoreProblem:
    """kcore_problem_t + kcore_enactor_t (gunrock/src/kcore/)."""

    def __init__(self, graph):
        h = C.c_void_p()
        check(lib.mgx_kcore_create(graph._h, C.byref(h)))
        self.graph, self._h = graph, h

    def reset(self):
        check(lib.mgx_kcore_reset(self._h))

    def enact(self):
        """-> (largest_k_core, {"rounds", "passes", "expanded", "removed"})"""
        largest = C.c_int()
        st = (C.c_int64 * 4)()
        check(lib.mgx_kcore_enact(self._h, C.byref(largest), st))
        return largest.value, {"rounds": st[0], "passes": st[1], "expanded": st[2], "removed": st[3]}

    def num_cores(self):
        out = np.empty(self.graph.num_nodes, dtype=np.int32)
        check(lib.mgx_kcore_num_cores(self._h, _ptr(out)))
        return out

    def degrees(self):
        out = np.empty(self.graph.num_nodes, dtype=np.int32)
        check(lib.mgx_kcore_degrees(self._h, _ptr(out)))
        return out

    def close(self):
        if self._h:
            lib.mgx_kcore_free(self._h)
            self._h = None


# building blocks ------------------------------------------------------------------------------
def scan_exclusive_i32(ctx, d_in, n, d_out):
    v = _i64()
    check(lib.mgx_scan_exclusive_i32(ctx._h, _dev_ptr(d_in), int(n), _dev_ptr(d_out), C.byref(v)))
    return v.value


def scan_frontier_degrees(graph, frontier, use_csc=False):
    v = _i64()
    check(lib.mgx_scan_frontier_degrees(graph._h, frontier._h, int(use_csc), C.byref(v)))
    return v.value


def lbs_expand_debug(graph, frontier, total):
    seg = np.empty(max(total, 1), dtype=np.int32)
    rank = np.empty(max(total, 1), dtype=np.int32)
    check(lib.mgx_lbs_expand_debug(graph._h, frontier._h, int(total), _ptr(seg), _ptr(rank)))
    return seg[:total], rank[:total]


def compact_i32(ctx, d_in, n, drop_value, d_out):
    v = _i64()
    check(lib.mgx_compact_i32(ctx._h, _dev_ptr(d_in), int(n), int(drop_value), _dev_ptr(d_out), C.byref(v)))
    return v.value


def segreduce(graph, frontier, d_vertex_value, identity, d_reduced, op="f32_plus", push=True):
    v = _i64()
    if op == "f32_plus":
        check(lib.mgx_segreduce_f32_plus(graph._h, frontier._h, int(push), _dev_ptr(d_vertex_value),
                                         C.c_float(identity), _dev_ptr(d_reduced), C.byref(v)))
    elif op == "i32_min":
        check(lib.mgx_segreduce_i32_min(graph._h, frontier._h, int(push), _dev_ptr(d_vertex_value), int(identity),
                                        _dev_ptr(d_reduced), C.byref(v)))
    elif op == "i32_max":
        check(lib.mgx_segreduce_i32_max(graph._h, frontier._h, int(push), _dev_ptr(d_vertex_value), int(identity),
                                        _dev_ptr(d_reduced), C.byref(v)))
    else:
        raise ValueError(op)
    return v.value


def rmat_edges(ctx, scale, first_edge, count, seed, scramble, d_src, d_dst, d_weight=None):
    check(lib.mgx_rmat_edges(ctx._h, int(scale), int(first_edge), int(count), C.c_uint64(seed), int(bool(scramble)),
                             _dev_ptr(d_src), _dev_ptr(d_dst), _dev_ptr(d_weight)))
