"""Vertex-range partitioned SSSP across the GPUs of one node ("(dst, dist) pairs with min-combining").

One process per GPU, the partition of mini_amd.dist_bfs (generation 1): rank r owns the global ids [r*chunk, (r+1)*chunk) --
their CSR rows (local row_offsets, GLOBAL col_indices, weights) and distances.  A superstep is frontier Bellman-Ford, the
reference's SSSP loop (gunrock/src/sssp/sssp_enactor.hxx:40-72):
    expand   (device)  relax the local frontier's edges; remote targets -> per-owner bins of 8-byte pairs
                       (vertex << 32 | float bits), one pair per target vertex and superstep holding the minimum over all
                       of this rank's edges to it (min-combining before send), and only when it beats every pair this rank
                       sent for the vertex before
    exchange (RCCL)    all_to_all of the bin sizes, then all_to_all_v of the pairs (int64): on the xGMI full mesh every
                       pair of GPUs has its own link
    receive  (device)  the owner keeps the minimum; vertices whose distance dropped form the next local frontier
    all_reduce(1 int)  global next-frontier size -> termination
DistSssp only needs an `engine` with reset/expand/send_bin/receive/swap/distances; HipSsspRankEngine is the product's
(C-ABI mgx_dsssp_*); the gloo CPU tests plug a small numpy engine to exercise this exchange/termination logic without a GPU.
"""
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from ._lib import check, lib
from .dist_bfs import NativeComm, chunk_of, range_of


class HipSsspRankEngine:
    """The per-rank device pieces (include/mgx/sssp_dist.hpp) behind the C-ABI."""

    def __init__(self, ctx, n_global, ranks, rank, row_offsets_local, col_indices_global, weights):
        self.ctx, self.n_global, self.ranks, self.rank = ctx, n_global, ranks, rank
        self.lo, self.hi = range_of(n_global, ranks, rank)
        self._keep = (row_offsets_local, col_indices_global, weights)
        h = C.c_void_p()
        check(lib.mgx_dsssp_create(ctx._h, int(n_global), int(ranks), int(rank), int(col_indices_global.numel()),
                                   C.c_void_p(row_offsets_local.data_ptr()), C.c_void_p(col_indices_global.data_ptr()),
                                   C.c_void_p(weights.data_ptr()), C.byref(h)))
        self._h = h
        p, cap = C.c_void_p(), C.c_int64()
        check(lib.mgx_dsssp_bins(self._h, C.byref(p), C.byref(cap)))
        self.cap = cap.value
        self._bins_ptr = p.value
        self.device = row_offsets_local.device
        self._counts = [0] * ranks

    def reset(self, src):
        check(lib.mgx_dsssp_reset(self._h, int(src)))

    def expand(self):
        counts = (C.c_int64 * self.ranks)()
        edges = C.c_int64()
        check(lib.mgx_dsssp_expand(self._h, counts, C.byref(edges)))
        self._counts = [counts[r] for r in range(self.ranks)]
        return self._counts, edges.value

    def send_bin(self, r):
        """the pairs for rank r as an int64 device tensor (a view of the library's bin)"""
        n = self._counts[r]
        if n == 0:
            return torch.empty(0, dtype=torch.int64, device=self.device)
        # the library's device memory, wrapped through the __cuda_array_interface__ protocol (no copy)
        holder = type("_DevArr", (), {"__cuda_array_interface__": {
            "shape": (n,), "typestr": "<i8", "data": (self._bins_ptr + 8 * r * self.cap, False), "version": 2}})()
        return torch.as_tensor(holder, device=self.device)

    def receive(self, pairs):
        if pairs.numel():
            check(lib.mgx_dsssp_receive(self._h, C.c_void_p(pairs.data_ptr()), int(pairs.numel())))

    def swap(self):
        v = C.c_int64()
        check(lib.mgx_dsssp_swap(self._h, C.byref(v)))
        return v.value

    def run_native(self, src, comm):
        """The whole superstep loop inside the library (mgx_dsssp_run) over its own RCCL communicator: no Python between
        the supersteps.  comm: mini_amd.dist_bfs.NativeComm or None (one rank)."""
        out = (C.c_int64 * 4)()
        check(lib.mgx_dsssp_run(self._h, comm._h if comm is not None else None, int(src), out))
        return {"iterations": out[0], "edges_local": out[1], "pairs_sent": out[2], "pairs_received": out[3]}

    def distances(self):
        out = np.empty(self.hi - self.lo, dtype=np.float32)
        check(lib.mgx_dsssp_distances(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def close(self):
        if self._h:
            lib.mgx_dsssp_free(self._h)
            self._h = None


class DistSssp:
    """Superstep driver.  comm_device: device the collectives run on ("cuda" for RCCL, "cpu" for gloo)."""

    def __init__(self, engine, rank, world, comm_device):
        self.e, self.rank, self.world, self.comm_device = engine, rank, world, torch.device(comm_device)
        # The loop inside the library (mgx_dsssp_run, direct RCCL calls on the context's stream) whenever the engine is the
        # HIP one and the collectives run on the GPUs; the Python loop below serves the gloo tests and MGX_DIST_NATIVE=0.
        # Same agreement protocol as DistBfs2: one rank without a communicator sends everybody to the Python loop.
        self.native, self.comm, self.native_error = False, None, None
        if hasattr(engine, "run_native") and self.comm_device.type == "cuda" and os.environ.get("MGX_DIST_NATIVE", "1") != "0":
            ok = 1
            if world > 1 or os.environ.get("MGX_DIST_FORCE_COLLECTIVES") == "1":
                try:
                    self.comm = NativeComm(engine.ctx, rank, world, self.comm_device)
                except Exception as ex:
                    self.comm, ok, self.native_error = None, 0, repr(ex)
                if world > 1:
                    flag = torch.tensor([ok], dtype=torch.int32, device=self.comm_device)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    ok = int(flag.item())
                    if not ok and self.comm is not None:
                        self.comm.close()
                        self.comm = None
            self.native = bool(ok)

    def _to_comm(self, t):
        return t if t.device == self.comm_device else t.to(self.comm_device)

    def run(self, src):
        e, W = self.e, self.world
        if self.native:
            return e.run_native(src, self.comm)
        e.reset(src)
        iterations, relaxed, sent = 0, 0, 0
        while True:
            counts, edges = e.expand()
            relaxed += edges
            if W > 1:
                send_counts = torch.tensor(counts, dtype=torch.int64, device=self.comm_device)
                send_counts[self.rank] = 0                      # (a rank never bins its own vertices: expand relaxed them)
                recv_counts = torch.empty(W, dtype=torch.int64, device=self.comm_device)
                dist.all_to_all_single(recv_counts, send_counts)
                in_splits = [int(c) if r != self.rank else 0 for r, c in enumerate(counts)]
                out_splits = [int(x) for x in recv_counts.tolist()]
                parts = [e.send_bin(r) for r in range(W) if r != self.rank and counts[r] > 0]
                send = torch.cat(parts) if parts else torch.empty(0, dtype=torch.int64, device=e.device)
                recv = torch.empty(sum(out_splits), dtype=torch.int64, device=self.comm_device)
                dist.all_to_all_single(recv, self._to_comm(send), out_splits, in_splits)
                sent += int(send.numel())
                if recv.numel():
                    e.receive(recv if recv.device == e.device else recv.to(e.device))
            nf = e.swap()
            iterations += 1
            if W > 1:
                t = torch.tensor([nf], dtype=torch.int64, device=self.comm_device)
                dist.all_reduce(t)
                nf = int(t.item())
            if nf == 0:
                break
        return {"iterations": iterations, "edges_local": relaxed, "pairs_sent": sent}

    def gather_distances(self):
        """Global distance array on every rank (validation only)."""
        loc = torch.from_numpy(self.e.distances())
        if self.world == 1:
            return loc.numpy()
        cap = chunk_of(self.e.n_global, self.world)
        pad = torch.full((cap,), -1.0, dtype=torch.float32)
        pad[: loc.numel()] = loc
        out = [torch.empty(cap, dtype=torch.float32, device=self.comm_device) for _ in range(self.world)]
        dist.all_gather(out, pad.to(self.comm_device))
        return torch.cat([o.cpu() for o in out])[: self.e.n_global].numpy()
