"""ctypes binding of oracle/liboracle.so -- the CPU checker.  Test infrastructure: only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg import this module."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_PATH = os.path.join(ROOT, "oracle", "liboracle.so")

_pi, _pf, _pi64 = C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_int64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t=C.c_int):
    return a.ctypes.data_as(C.POINTER(t))


class Oracle:
    def __init__(self, path=ORACLE_PATH):
        if not os.path.exists(path):
            raise FileNotFoundError("%s missing: run `make -C oracle`" % path)
        self.lib = L = C.CDLL(path)
        L.orc_scan_degrees.restype = C.c_int64
        L.orc_bfs_advance.restype = C.c_int64
        L.orc_bfs_filter.restype = C.c_int64
        L.orc_bfs_gen_unvisited.restype = C.c_int64
        L.orc_bfs_advance_backward.restype = C.c_int64
        L.orc_sssp_advance.restype = C.c_int64
        L.orc_sssp_filter.restype = C.c_int64
        L.orc_neighbor_reduce_f32_plus.restype = C.c_int64
        L.orc_neighbor_reduce_i32.restype = C.c_int64
        L.orc_rmat_scramble.restype = C.c_uint32
        L.orc_free.argtypes = [C.c_void_p]

    # ---- loader ----
    def load_mtx(self, path, undir=False, random_w=False):
        n, m = C.c_int(), C.c_int64()
        off, idx, w, src = _pi(), _pi(), _pf(), _pi()
        rc = self.lib.orc_load_mtx(str(path).encode(), int(undir), int(random_w), C.byref(n), C.byref(m),
                                   C.byref(off), C.byref(idx), C.byref(w), C.byref(src))
        if rc != 0:
            raise RuntimeError("orc_load_mtx rc=%d" % rc)
        N, M = n.value, m.value
        out = (N, np.ctypeslib.as_array(off, (N + 1,)).copy(),
               np.ctypeslib.as_array(idx, (max(M, 1),))[:M].copy(),
               np.ctypeslib.as_array(w, (max(M, 1),))[:M].copy(),
               np.ctypeslib.as_array(src, (max(M, 1),))[:M].copy())
        for p in (off, idx, w, src):
            self.lib.orc_free(p)
        return out

    def csr_from_tuples(self, n, t0, t1, w=None, undir=False):
        t0, t1 = _i32(t0), _i32(t1)
        wv = None if w is None else _f32(w)
        m = C.c_int64()
        off, idx, ww, src = _pi(), _pi(), _pf(), _pi()
        rc = self.lib.orc_csr_from_tuples(int(n), C.c_int64(len(t0)), _p(t0), _p(t1),
                                          None if wv is None else _p(wv, C.c_float), int(undir),
                                          C.byref(m), C.byref(off), C.byref(idx), C.byref(ww), C.byref(src))
        if rc != 0:
            raise RuntimeError("orc_csr_from_tuples rc=%d" % rc)
        M = m.value
        out = (np.ctypeslib.as_array(off, (n + 1,)).copy(), np.ctypeslib.as_array(idx, (max(M, 1),))[:M].copy(),
               np.ctypeslib.as_array(ww, (max(M, 1),))[:M].copy())
        for p in (off, idx, ww, src):
            self.lib.orc_free(p)
        return out

    # ---- reference CPU validators ----
    def bfs_cpu(self, ro, ci, src):
        ro, ci = _i32(ro), _i32(ci)
        labels = np.full(len(ro) - 1, -1, dtype=np.int32)
        self.lib.orc_bfs_cpu(len(ro) - 1, _p(ro), _p(ci), int(src), _p(labels))
        return labels

    def sssp_cpu(self, ro, ci, w, src):
        ro, ci, w = _i32(ro), _i32(ci), _f32(w)
        n = len(ro) - 1
        preds = np.full(n, -1, dtype=np.int32)
        dist = np.zeros(n, dtype=np.int32)
        self.lib.orc_sssp_cpu(n, _p(ro), _p(ci), _p(w, C.c_float), int(src), _p(preds), _p(dist))
        return preds, dist

    # ---- primitives ----
    def scan_degrees(self, offsets, ids):
        offsets, ids = _i32(offsets), _i32(ids)
        out = np.zeros(max(len(ids), 1), dtype=np.int32)
        tot = self.lib.orc_scan_degrees(_p(offsets), _p(ids), C.c_int64(len(ids)), _p(out))
        return out[:len(ids)], tot

    def lbs(self, scanned, total):
        scanned = _i32(scanned)
        seg = np.zeros(max(total, 1), dtype=np.int32)
        rank = np.zeros(max(total, 1), dtype=np.int32)
        self.lib.orc_lbs(_p(scanned), C.c_int64(len(scanned)), C.c_int64(total), _p(seg), _p(rank))
        return seg[:total], rank[:total]

    # ---- BFS operators ----
    def bfs_advance(self, ro, ci, labels, fin, iteration):
        ro, ci, fin = _i32(ro), _i32(ci), _i32(fin)
        _, tot = self.scan_degrees(ro, fin)
        out = np.zeros(max(tot, 1), dtype=np.int32)
        front = self.lib.orc_bfs_advance(_p(ro), _p(ci), _p(labels), _p(fin), C.c_int64(len(fin)), int(iteration),
                                         _p(out))
        return out[:front]

    def bfs_filter(self, fin):
        fin = _i32(fin)
        out = np.zeros(max(len(fin), 1), dtype=np.int32)
        k = self.lib.orc_bfs_filter(_p(fin), C.c_int64(len(fin)), _p(out))
        return out[:k]

    # the pull-direction operators (advance.hxx:69-160), one call each
    def bfs_gen_unvisited(self, labels, indices):
        labels, indices = _i32(labels), _i32(indices)
        out = np.zeros(max(len(indices), 1), dtype=np.int32)
        k = self.lib.orc_bfs_gen_unvisited(_p(labels), _p(indices), C.c_int64(len(indices)), _p(out))
        return out[:k]

    def bfs_sparse_to_dense(self, labels, sparse, dense, iteration):
        """writes dense[v] for the listed vertices only, in place (what the operator does)"""
        labels, sparse = _i32(labels), _i32(sparse)
        assert dense.dtype == np.int32 and dense.flags.c_contiguous
        self.lib.orc_bfs_sparse_to_dense(_p(labels), _p(sparse), C.c_int64(len(sparse)), _p(dense), int(iteration))

    def bfs_advance_backward(self, co, ri, labels, unvisited, bitmap, bitmap_out, iteration):
        """one bottom-up step, in place on labels / unvisited (claimed slots become -1) / bitmap_out; returns the
        in-edges inspected"""
        co, ri, bitmap = _i32(co), _i32(ri), _i32(bitmap)
        for a in (labels, unvisited, bitmap_out):
            assert a.dtype == np.int32 and a.flags.c_contiguous
        return self.lib.orc_bfs_advance_backward(_p(co), _p(ri), _p(labels), _p(unvisited), C.c_int64(len(unvisited)),
                                                 _p(bitmap), _p(bitmap_out), int(iteration))

    def bfs_enact_pushpull(self, ro, ci, src, threshold, co=None, ri=None):
        ro, ci = _i32(ro), _i32(ci)
        co = ro if co is None else _i32(co)
        ri = ci if ri is None else _i32(ri)
        n = len(ro) - 1
        labels = np.zeros(n, dtype=np.int32)
        stats = np.zeros(4, dtype=np.int64)
        rc = self.lib.orc_bfs_enact_pushpull(n, C.c_int64(len(ci)), _p(ro), _p(ci), _p(co), _p(ri), int(src),
                                             C.c_float(threshold), _p(labels), _p(stats, C.c_int64))
        return rc, labels, stats

    # ---- SSSP ----
    def sssp_enact(self, ro, ci, w, src, queue_sizing=1.0):
        ro, ci, w = _i32(ro), _i32(ci), _f32(w)
        n = len(ro) - 1
        dist = np.zeros(n, dtype=np.float32)
        preds = np.zeros(n, dtype=np.int32)
        stats = np.zeros(3, dtype=np.int64)
        rc = self.lib.orc_sssp_enact(n, C.c_int64(len(ci)), _p(ro), _p(ci), _p(w, C.c_float), int(src),
                                     C.c_float(queue_sizing), _p(dist, C.c_float), _p(preds), _p(stats, C.c_int64))
        if rc != 0:
            raise RuntimeError("orc_sssp_enact rc=%d" % rc)
        return dist, preds, stats

    def sssp_dijkstra_f32(self, ro, ci, w, src):
        ro, ci, w = _i32(ro), _i32(ci), _f32(w)
        n = len(ro) - 1
        dist = np.zeros(n, dtype=np.float32)
        self.lib.orc_sssp_dijkstra_f32(n, _p(ro), _p(ci), _p(w, C.c_float), int(src), _p(dist, C.c_float))
        return dist

    # ---- neighbour reduce / PR ----
    def neighbor_reduce_f32_plus(self, off, idx, fin, values, identity=0.0):
        off, idx, fin, values = _i32(off), _i32(idx), _i32(fin), _f32(values)
        red = np.zeros(max(len(fin), 1), dtype=np.float32)
        nz = self.lib.orc_neighbor_reduce_f32_plus(_p(off), _p(idx), _p(fin), C.c_int64(len(fin)),
                                                   _p(values, C.c_float), C.c_float(identity), _p(red, C.c_float))
        return red[:len(fin)], nz

    def neighbor_reduce_i32(self, off, idx, fin, values, identity, is_max):
        off, idx, fin, values = _i32(off), _i32(idx), _i32(fin), _i32(values)
        red = np.zeros(max(len(fin), 1), dtype=np.int32)
        nz = self.lib.orc_neighbor_reduce_i32(_p(off), _p(idx), _p(fin), C.c_int64(len(fin)), _p(values),
                                              int(identity), int(is_max), _p(red))
        return red[:len(fin)], nz

    def pr_enact(self, off, idx, max_iter):
        off, idx = _i32(off), _i32(idx)
        n = len(off) - 1
        ranks = np.zeros(n, dtype=np.float32)
        lens = np.zeros(max(max_iter, 1), dtype=np.int64)
        it = self.lib.orc_pr_enact(n, _p(off), _p(idx), int(max_iter), _p(ranks, C.c_float), _p(lens, C.c_int64))
        return ranks, lens[:it]

    # ---- k-core ----
    def kcore_cpu(self, off, idx):
        """kcore_problem_t::cpu: (core numbers, largest_k_core)"""
        off, idx = _i32(off), _i32(idx)
        n = len(off) - 1
        cores = np.zeros(max(n, 1), dtype=np.int32)
        largest = self.lib.orc_kcore_cpu(n, _p(off), _p(idx), _p(cores))
        return cores[:n], largest

    def kcore_enact(self, off, idx):
        """kcore_enactor_t::enact over serial operators: (core numbers, largest_k_core, stats[4])"""
        off, idx = _i32(off), _i32(idx)
        n = len(off) - 1
        cores = np.zeros(max(n, 1), dtype=np.int32)
        stats = np.zeros(4, dtype=np.int64)
        largest = self.lib.orc_kcore_enact(n, _p(off), _p(idx), _p(cores), _p(stats, C.c_int64))
        return cores[:n], largest, stats

    # ---- RMAT spec ----
    def rmat_edges(self, scale, first, count, seed, scramble=True, weighted=True):
        s = np.zeros(max(count, 1), dtype=np.int32)
        d = np.zeros(max(count, 1), dtype=np.int32)
        w = np.zeros(max(count, 1), dtype=np.float32)
        self.lib.orc_rmat_edges(int(scale), C.c_int64(first), C.c_int64(count), C.c_uint64(seed), int(scramble),
                                _p(s), _p(d), _p(w, C.c_float) if weighted else None)
        return s[:count], d[:count], w[:count]

    def rmat_csr(self, scale, edgefactor, seed, scramble=True, undir=True):
        n = 1 << scale
        s, d, w = self.rmat_edges(scale, 0, edgefactor * n, seed, scramble)
        ro, ci, ww = self.csr_from_tuples(n, s, d, w, undir)
        return n, ro, ci, ww
