"""Vertex-range partitioned BFS across the GPUs of one node (BASELINE config 5).

One process per GPU.  Rank r owns the global ids [r*chunk, (r+1)*chunk) -- their CSR rows (local
row_offsets, GLOBAL col_indices) and labels.  A superstep is
    expand   (device)  local frontier -> per-owner bins of neighbour ids, each id sent at most once
                       per traversal per rank (rank-private `seen` bitmap = the pre-send dedup)
    exchange (RCCL)    all_to_all of the bin sizes, then all_to_all_v of the ids: on MI355X's xGMI full
                       mesh every pair of GPUs has its own link, so all 7 links of a GPU are busy at
                       once; nothing is ring-reduced
    receive  (device)  owner labels what it had not labelled and appends it to its next frontier
    all_reduce(1 int)  global next-frontier size -> termination
DistBfs only needs an `engine` with reset/expand/send_bin/receive/swap/labels; HipRankEngine is the
product's (C-ABI mgx_dbfs_*); the gloo CPU test plugs a small numpy engine to exercise this exact
exchange/termination logic without a GPU.
"""
import ctypes as C
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from . import api
from ._lib import check, lib


def chunk_of(n_global, ranks):
    return (n_global + ranks - 1) // ranks


def range_of(n_global, ranks, rank):
    c = chunk_of(n_global, ranks)
    return min(rank * c, n_global), min((rank + 1) * c, n_global)


class HipRankEngine:
    """The per-rank device pieces (include/mgx/bfs_dist.hpp) behind the C-ABI."""

    def __init__(self, ctx, n_global, ranks, rank, row_offsets_local, col_indices_global):
        self.ctx, self.n_global, self.ranks, self.rank = ctx, n_global, ranks, rank
        self.lo, self.hi = range_of(n_global, ranks, rank)
        self.cap = chunk_of(n_global, ranks)
        self._keep = (row_offsets_local, col_indices_global)
        self.bins = torch.empty(ranks * self.cap, dtype=torch.int32, device=row_offsets_local.device)
        h = C.c_void_p()
        check(lib.mgx_dbfs_create(ctx._h, int(n_global), int(ranks), int(rank), int(col_indices_global.numel()),
                                  C.c_void_p(row_offsets_local.data_ptr()),
                                  C.c_void_p(col_indices_global.data_ptr()),
                                  C.c_void_p(self.bins.data_ptr()), int(self.cap), C.byref(h)))
        self._h = h
        self._counts = [0] * ranks

    def reset(self, src):
        check(lib.mgx_dbfs_reset(self._h, int(src)))

    def expand(self):
        counts = (C.c_int64 * self.ranks)()
        edges = C.c_int64()
        check(lib.mgx_dbfs_expand(self._h, counts, C.byref(edges)))
        self._counts = [counts[r] for r in range(self.ranks)]
        return self._counts, edges.value

    def send_bin(self, r):
        return self.bins[r * self.cap: r * self.cap + self._counts[r]]

    def receive(self, ids, label):
        if ids.numel():
            check(lib.mgx_dbfs_receive(self._h, C.c_void_p(ids.data_ptr()), int(ids.numel()), int(label)))

    def swap(self):
        v = C.c_int64()
        check(lib.mgx_dbfs_swap(self._h, C.byref(v)))
        return v.value

    def labels(self):
        out = np.empty(self.hi - self.lo, dtype=np.int32)
        check(lib.mgx_dbfs_labels(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def close(self):
        if self._h:
            lib.mgx_dbfs_free(self._h)
            self._h = None


class DistBfs:
    """Superstep driver.  comm_device: device the collectives run on ("cuda" for RCCL, "cpu" for gloo)."""

    def __init__(self, engine, rank, world, comm_device):
        self.e, self.rank, self.world, self.comm_device = engine, rank, world, torch.device(comm_device)
        self.expand_seconds = 0.0

    def _to_comm(self, t):
        return t if t.device == self.comm_device else t.to(self.comm_device)

    def run(self, src):
        e, W = self.e, self.world
        e.reset(src)
        level, edges_local = 0, 0
        self.levels_edges = []
        while True:
            counts, edges = e.expand()
            edges_local += edges
            self.levels_edges.append(edges)
            own = e.send_bin(self.rank)
            if W > 1:
                send_counts = torch.tensor(counts, dtype=torch.int64, device=self.comm_device)
                send_counts[self.rank] = 0                      # the own bin never leaves the device
                recv_counts = torch.empty(W, dtype=torch.int64, device=self.comm_device)
                dist.all_to_all_single(recv_counts, send_counts)
                in_splits = [int(c) if r != self.rank else 0 for r, c in enumerate(counts)]
                out_splits = [int(x) for x in recv_counts.tolist()]
                parts = [e.send_bin(r) for r in range(W) if r != self.rank and counts[r] > 0]
                dev = own.device
                send = torch.cat(parts) if parts else torch.empty(0, dtype=torch.int32, device=dev)
                recv = torch.empty(sum(out_splits), dtype=torch.int32, device=self.comm_device)
                dist.all_to_all_single(recv, self._to_comm(send), out_splits, in_splits)
                recv = recv if recv.device == dev else recv.to(dev)
            else:
                recv = None
            e.receive(own, level + 1)
            if recv is not None:
                e.receive(recv, level + 1)
            nf = e.swap()
            level += 1
            if W > 1:
                t = torch.tensor([nf], dtype=torch.int64, device=self.comm_device)
                dist.all_reduce(t)
                nf = int(t.item())
            if nf == 0:
                break
        self.levels = level
        return {"levels": level, "edges_local": edges_local}

    def gather_labels(self):
        """Global label array on every rank (validation only)."""
        loc = torch.from_numpy(self.e.labels())
        if self.world == 1:
            return loc.numpy()
        cap = chunk_of(self.e.n_global, self.world)
        pad = torch.full((cap,), -2, dtype=torch.int32)
        pad[: loc.numel()] = loc
        out = [torch.empty(cap, dtype=torch.int32, device=self.comm_device) for _ in range(self.world)]
        dist.all_gather(out, pad.to(self.comm_device))
        full = torch.cat([o.cpu() for o in out])[: self.e.n_global]
        return full.numpy()


# ---- shard construction ---------------------------------------------------------------------------
def rmat_shard_csr(ctx, scale, edgefactor, seed, ranks, rank, device, pairs_per_chunk=1 << 26):
    """Local CSR rows [lo,hi) of the symmetrised R-MAT graph, GLOBAL neighbour ids.

    Every rank streams the whole (scale, seed) pair stream through the device generator in chunks and
    keeps the entries whose ROW falls in its range (row = dst for the pair, row = src for the swapped
    copy, graph.hxx:130-137) -- no host ever holds the 2^31-entry graph.
    """
    n = 1 << scale
    lo, hi = range_of(n, ranks, rank)
    total = edgefactor * n
    rows_l, nbrs_l = [], []
    for first in range(0, total, pairs_per_chunk):
        cnt = min(pairs_per_chunk, total - first)
        s = torch.empty(cnt, dtype=torch.int32, device=device)
        d = torch.empty(cnt, dtype=torch.int32, device=device)
        torch.cuda.synchronize(device)
        api.rmat_edges(ctx, scale, first, cnt, seed, True, s, d, None)
        ctx.synchronize()
        m1 = (d >= lo) & (d < hi)            # pair (u,v): row v, neighbour u
        rows_l.append(d[m1]); nbrs_l.append(s[m1])
        m2 = (s >= lo) & (s < hi)            # swapped copy: row u, neighbour v
        rows_l.append(s[m2]); nbrs_l.append(d[m2])
        del s, d, m1, m2
    rows = torch.cat(rows_l); nbrs = torch.cat(nbrs_l)
    del rows_l, nbrs_l
    key = ((rows.to(torch.int64) - lo) << 32) | nbrs.to(torch.int64)
    del rows, nbrs
    key, _ = torch.sort(key)
    col = (key & 0xFFFFFFFF).to(torch.int32)
    lrow = key >> 32
    del key
    counts = torch.bincount(lrow, minlength=hi - lo)
    ro = torch.zeros(hi - lo + 1, dtype=torch.int64, device=device)
    torch.cumsum(counts, 0, out=ro[1:])
    return ro.to(torch.int32), col


def pick_sources_dist(row_offsets_local, lo, hi, n_global, count, seed, comm_device):
    """Same source list on every rank: splitmix64(seed+i) mod n, skipping degree-0 vertices (owner decides)."""
    from .rmat import _mix64_py
    cand = [int(_mix64_py(seed + i) % n_global) for i in range(8 * count + 64)]
    flags = torch.zeros(len(cand), dtype=torch.int64)
    ro = row_offsets_local
    for j, v in enumerate(cand):
        if lo <= v < hi and int(ro[v - lo + 1]) > int(ro[v - lo]):
            flags[j] = 1
    if dist.is_initialized() and dist.get_world_size() > 1:
        f = flags.to(comm_device)
        dist.all_reduce(f)
        flags = f.cpu()
    return [v for v, ok in zip(cand, flags.tolist()) if ok][:count]


def bench_main(args, rank, world, local_rank):
    """bench.py body for N > 1 (one process per GPU, RCCL).  Weak scaling: per-GPU graph share fixed,
    global scale = --scale + log2(N) (Graph500-style)."""
    import json
    import mini_amd
    device = torch.device("cuda", local_rank)
    stream = torch.cuda.current_stream()
    ctx = mini_amd.Context(local_rank, stream.cuda_stream)
    gscale = args.scale + int(np.log2(world))
    seed = gscale if args.seed is None else args.seed
    n = 1 << gscale
    t_build = time.time()
    ro, col = rmat_shard_csr(ctx, gscale, args.edgefactor, seed, world, rank, device)
    torch.cuda.synchronize()
    t_build = time.time() - t_build
    eng = HipRankEngine(ctx, n, world, rank, ro, col)
    bfs = DistBfs(eng, rank, world, "cuda")
    ro_host = ro.cpu().numpy()
    sources = pick_sources_dist(ro_host, eng.lo, eng.hi, n, args.steps + args.warmup, seed, device)
    for s in sources[: args.warmup]:
        bfs.run(s)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    edges_local, levels = 0, 0
    for s in sources[args.warmup:]:
        st = bfs.run(s)
        edges_local += st["edges_local"]
        levels += st["levels"]
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    e = torch.tensor([edges_local], dtype=torch.int64, device=device)
    dist.all_reduce(e)
    m_t = int(e.item())
    if rank == 0:
        value = m_t / elapsed / 1e6
        out = {"metric": "MTEPS (million traversed edges/sec) BFS advance+filter, RMAT-%d" % gscale,
               "value": round(value, 2), "unit": "MTEPS", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / max(args.steps, 1), 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
               "data": "synthetic",
               "config": {"workload": "BFS push on RMAT scale %d (= %d per GPU + log2 N) ef %d, symmetrised, "
                                      "1-D vertex-range partition over %d GPUs, all-to-all frontier exchange (RCCL), "
                                      "%d seeded sources" % (gscale, args.scale, args.edgefactor, world, args.steps),
                          "scale": gscale, "edgefactor": args.edgefactor, "seed": seed,
                          "parallelism": "vertex-range x%d" % world},
               "roofline": {"bound": "hbm", "kernel": "k_transform_lbs (dist expand)",
                            "achieved": round(8.0 * m_t / world / elapsed / 1e9, 2), "peak": 8000.0, "unit": "GB/s",
                            "frac": round(8.0 * m_t / world / elapsed / 1e9 / 8000.0, 5), "traffic": None,
                            "note": "per-GPU algorithmic bytes (8 B/edge) over the whole superstep loop incl. exchange"},
               "cpu_baseline": None, "avg_levels": round(levels / max(args.steps, 1), 2),
               "graph_build_s": round(t_build, 2)}
        print(json.dumps(out), flush=True)
    dist.barrier()
    dist.destroy_process_group()
