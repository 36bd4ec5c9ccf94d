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
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

from . import api
from ._lib import MgxError, check, lib


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
        self.levels = level + 1          # levels 0..level hold vertices (reference "iterations")
        return {"levels": level + 1, "edges_local": edges_local}

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


# =====================================================================================================
# Generation 2: fused kernels per rank + bitmap exchange (include/mgx/bfs_dist2.hpp)
# =====================================================================================================
def local_rows(n_global, ranks, rank):
    """number of vertices owned by `rank` under the cyclic partition v % ranks"""
    return (n_global - rank + ranks - 1) // ranks


def bitmap_words(n_global):
    """32-bit words of a generation-2 bitmap (padded to 16 bytes; == mgx_dbfs2_words)"""
    return ((n_global + 31) // 32 + 3) // 4 * 4


def list_words(n_global, ranks):
    """32-bit words of a rank's id list on sparse levels: 4 header words ([0] = count) + n / (256 ranks) ids, at least 252
    (== mgx_dbfs2_list_words)"""
    c = max(n_global // (256 * ranks), 252)
    return (c + 4 + 3) // 4 * 4


def exchange_words(n_global, ranks):
    """length of the buffer a rank's new-bit map is exchanged in: bitmap_words padded so that it splits into
    `ranks` slices of whole 16-byte units (the reduce-scatter exchange sends slice r to rank r)"""
    unit = 4 * ranks
    return (bitmap_words(n_global) + unit - 1) // unit * unit


class HipRankEngine2:
    """Per-rank device side of the bitmap-exchange BFS (C-ABI mgx_dbfs2_*).  Ids are global, hub-first.
    reset / push / merge only enqueue work on the context's stream; status() synchronises."""

    def __init__(self, ctx, n_global, ranks, rank, row_offsets_local, col_indices_global):
        self.ctx, self.n_global, self.ranks, self.rank = ctx, n_global, ranks, rank
        self.n_local = local_rows(n_global, ranks, rank)
        w = C.c_int64()
        check(lib.mgx_dbfs2_words(int(n_global), C.byref(w)))
        self.nwords = w.value
        assert self.nwords == bitmap_words(n_global)
        self._keep = (row_offsets_local, col_indices_global)
        # (the words past nwords are never written: they stay zero in every exchange)
        self.newbits = torch.zeros(exchange_words(n_global, ranks), dtype=torch.int32, device=row_offsets_local.device)
        h = C.c_void_p()
        check(lib.mgx_dbfs2_create(ctx._h, int(n_global), int(ranks), int(rank),
                                   C.c_void_p(row_offsets_local.data_ptr()), C.c_void_p(col_indices_global.data_ptr()),
                                   C.c_void_p(self.newbits.data_ptr()), C.byref(h)))
        self._h = h
        # unit blocks of the rank's long rows: big levels read them instead of walking the long-row queue (MGX_DIST_UNITS=0: off)
        self.units = 0
        if os.environ.get("MGX_DIST_UNITS", "1") != "0":
            u = C.c_int64()
            check(lib.mgx_dbfs2_build_units(self._h, C.byref(u)))
            self.units = u.value
        # id lists of the sparse levels (MGX_DIST_LISTS=0: bitmaps on every level)
        self.list = None
        if os.environ.get("MGX_DIST_LISTS", "1") != "0":
            w = C.c_int64()
            check(lib.mgx_dbfs2_list_words(int(n_global), int(ranks), C.byref(w)))
            assert w.value == list_words(n_global, ranks)
            self.list = torch.zeros(w.value, dtype=torch.int32, device=row_offsets_local.device)
            check(lib.mgx_dbfs2_set_list(self._h, C.c_void_p(self.list.data_ptr()), w.value))

    def dense_levels(self):
        """levels of the last traversal that read the long rows from the unit blocks (valid after status() / run_native())"""
        v = C.c_int64()
        check(lib.mgx_dbfs2_dense_levels(self._h, C.byref(v)))
        return v.value

    def cold_levels(self):
        """(levels of the last traversal that ran the cold-edge pass, pairs in the rank's cold lists)"""
        v, p = C.c_int64(), C.c_int64()
        check(lib.mgx_dbfs2_cold_levels(self._h, C.byref(v), C.byref(p)))
        return v.value, p.value

    def path_levels(self):
        """(levels appended by the push, levels with vertex-by-vertex short rows, 1 if a list was declared overflowed, packed
        cold-edge slices) of the last traversal, valid after status() / run_native()"""
        o = (C.c_int64 * 4)()
        check(lib.mgx_dbfs2_path_levels(self._h, o))
        return tuple(int(v) for v in o)

    def reset(self, src):
        check(lib.mgx_dbfs2_reset(self._h, int(src)))

    def push(self, level):
        """enqueues the level's push; returns the rank's new-bit map (and fills self.list, if lists are on)"""
        check(lib.mgx_dbfs2_push(self._h, int(level)))
        return self.newbits

    def apply_lists(self, level, lists, nlists):
        """lists: nlists gathered id lists -> (overflow: exchange the bitmaps, sum of the counts: 0 = traversal over)"""
        o = (C.c_int64 * 3)()
        check(lib.mgx_dbfs2_apply_lists(self._h, int(level), C.c_void_p(lists.data_ptr()), int(nlists),
                                        int(lists.numel() // nlists), o))
        return bool(o[0]), int(o[1])

    def merge(self, level, maps, nmaps):
        """maps: nmaps new-bit maps, each as long as the buffer push() returned"""
        check(lib.mgx_dbfs2_merge_maps(self._h, int(level), C.c_void_p(maps.data_ptr()), int(nmaps),
                                       int(maps.numel() // nmaps)))

    def or_maps(self, maps, nmaps, out):
        """out[w] = OR of the nmaps equally long maps (device tensors on this engine's GPU; out may alias maps)"""
        words = maps.numel() // nmaps
        check(lib.mgx_dbfs2_or_maps(self._h, C.c_void_p(maps.data_ptr()), int(nmaps), int(words), int(words),
                                    C.c_void_p(out.data_ptr())))

    def status(self, next_level):
        o = (C.c_int64 * 6)()
        check(lib.mgx_dbfs2_status(self._h, int(next_level), o))
        return {"over": bool(o[0]), "levels": o[1], "edges_local": o[2], "new_global": o[3]}

    def run_native(self, src, comm, exchange):
        """a whole traversal inside the library (mgx_dbfs2_run): push -> RCCL exchange -> merge per level, enqueued in
        batches from C++ on the context's stream -- no Python between levels.  comm: NativeComm or None (one rank)."""
        o = (C.c_int64 * 6)()
        check(lib.mgx_dbfs2_run(self._h, comm._h if comm is not None else None, int(src), 1 if exchange == "reduce" else 0,
                                int(self.newbits.numel()), o))
        return {"over": bool(o[0]), "levels": o[1], "edges_local": o[2], "new_global": o[3]}

    def spec_stats(self):
        """(traversals planned ahead, frozen by a list that overflowed against the plan, longer than planned, levels of the last
        plan, its lists-mask) -- mgx_dbfs2_spec_stats"""
        o = (C.c_int64 * 5)()
        check(lib.mgx_dbfs2_spec_stats(self._h, o))
        return tuple(int(v) for v in o)

    def forget_plan(self):
        """drop this engine's traversal history: its next plan is "none", as a recreated handle's (mgx_dbfs2_forget_plan)"""
        check(lib.mgx_dbfs2_forget_plan(self._h))

    @staticmethod
    def run_group(engines, src):
        """all ranks' engines (made on ONE context) in turn from this thread, collectives as device copies (mgx_dbfs2_run_group)"""
        G = len(engines)
        hs = (C.c_void_p * G)(*[e._h for e in engines])
        o = (C.c_int64 * (6 * G))()
        check(lib.mgx_dbfs2_run_group(hs, G, int(src), int(engines[0].newbits.numel()), o))
        return [{"over": bool(o[6 * r]), "levels": o[6 * r + 1], "edges_local": o[6 * r + 2], "new_global": o[6 * r + 3]} for r in range(G)]

    def labels(self):
        out = np.empty(self.n_local, dtype=np.int32)
        check(lib.mgx_dbfs2_labels(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def visited(self):
        """this rank's copy of the visited bitmap over all vertices (uint32 words; bit v = vertex v, hub-first global ids)"""
        out = np.empty(self.nwords, dtype=np.uint32)
        check(lib.mgx_dbfs2_visited(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def close(self):
        if self._h:
            lib.mgx_dbfs2_free(self._h)
            self._h = None


def _lib_status_invalid():
    from ._lib import MGX_E_INVALID
    return MGX_E_INVALID


class NativeComm:
    """An RCCL communicator owned by the library (include/mgx/comm.hpp): rank 0 makes the unique id, torch.distributed
    carries its 128 bytes to the other ranks, every rank joins (ncclCommInitRank)."""

    def __init__(self, ctx, rank, world, comm_device):
        # The collective sequence is the same on every rank whatever fails where: rank 0 ALWAYS
        # broadcasts 129 bytes -- the id and an ok flag; if it could not make an id (librccl not loadable, a symbol
        # missing) the flag is 0 and every rank raises the same error after the broadcast, instead of rank 0 skipping the
        # broadcast and the others waiting in it.
        self._h = None
        # (ncclCommInitRank is itself a collective: a rank that cannot load RCCL must not leave the others waiting in it)
        avail = torch.tensor([int(lib.mgx_comm_available())], dtype=torch.int32, device=comm_device)
        if world > 1:
            dist.all_reduce(avail, op=dist.ReduceOp.MIN)
        if int(avail.item()) != 1:
            raise MgxError(_lib_status_invalid(), "RCCL is not loadable on every rank (mgx_comm_available)")
        ident = torch.zeros(129, dtype=torch.uint8)
        err = None
        if rank == 0:
            buf = (C.c_ubyte * 128)()
            try:
                check(lib.mgx_comm_unique_id(buf))
                ident[:128] = torch.tensor(list(buf), dtype=torch.uint8)
                ident[128] = 1
            except Exception as ex:
                err = ex
        if world > 1:
            t = ident.to(comm_device)
            dist.broadcast(t, 0)
            ident = t.cpu()
        if int(ident[128]) != 1:
            raise err if err is not None else MgxError(_lib_status_invalid(), "rank 0 could not create an RCCL unique id")
        raw = (C.c_ubyte * 128)(*ident[:128].tolist())
        h = C.c_void_p()
        check(lib.mgx_comm_create(ctx._h, int(world), int(rank), raw, C.byref(h)))
        self._h = h
        self.library = (lib.mgx_comm_library() or b"").decode()

    def close(self):
        if self._h:
            lib.mgx_comm_free(self._h)
            self._h = None


class LoopbackComm:
    """A communicator of the library's in-process stand-in for RCCL (include/mgx/comm_loopback.hpp): the ranks are host THREADS
    of this process.  One thread makes the id (LoopbackComm.new_id()), every rank thread constructs its communicator from it --
    the constructor blocks until all `world` threads have joined, as ncclCommInitRank does.  What the -m gpu tests run
    mgx_dbfs2_run / mgx_dsssp_run over with 2 .. 8 ranks on one GPU; never selected by a product path."""

    @staticmethod
    def new_id():
        buf = (C.c_ubyte * 128)()
        check(lib.mgx_comm_loopback_id(buf))
        return bytes(buf)

    def __init__(self, ctx, rank, world, ident):
        self._h = None
        raw = (C.c_ubyte * 128)(*ident)
        h = C.c_void_p()
        check(lib.mgx_comm_create(ctx._h, int(world), int(rank), raw, C.byref(h)))
        self._h = h
        self.library = "loopback"

    def rounds(self):
        """collective rounds the world has completed (one per collective outside a group, one per group)"""
        loop, rounds = C.c_int(), C.c_int64()
        check(lib.mgx_comm_info(self._h, C.byref(loop), C.byref(rounds)))
        assert loop.value == 1
        return rounds.value

    def close(self):
        if self._h:
            lib.mgx_comm_free(self._h)
            self._h = None


def run_rank_threads(world, fn, timeout_s=300.0):
    """fn(rank) on `world` host threads at once (ctypes calls release the GIL: the ranks really run side by side); returns the
    list of results by rank, re-raises the first rank's exception.  A rank that never returns is reported, not waited for."""
    import threading
    out, err = [None] * world, [None] * world

    def body(r):
        try:
            out[r] = fn(r)
        except BaseException as ex:      # noqa: BLE001 -- handed to the caller
            err[r] = ex

    ts = [threading.Thread(target=body, args=(r,), daemon=True) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout_s)
    hung = [r for r, t in enumerate(ts) if t.is_alive()]
    if hung:
        raise RuntimeError("rank threads %r did not return within %.0f s" % (hung, timeout_s))
    for r, ex in enumerate(err):
        if ex is not None:
            raise ex
    return out


class DistBfs2:
    """Superstep driver of generation 2: push (device) -> exchange of the new-bit maps -> merge (device), all
    stream-ordered.  The exchange is an OR-all-reduce of bitmaps, which RCCL does not have as such:
      "gather": one all_gather, every rank ORs the R maps itself -- R-1 bitmaps received per rank and level;
      "reduce": all_to_all of slices (rank r gets slice r of every map and ORs them), then all_gather of the merged
                slices -- 2 (R-1)/R bitmaps per rank and level, two collectives.
    Default: "reduce" from 4 ranks on (MGX_DIST_EXCHANGE overrides).  Every rank counts the merged discoveries
    itself, so there is no reduction for the termination test; the host looks at the device state once per BATCH
    of levels: the first batch is as long as the
    previous traversal was (sources differ, the level structure of a graph hardly), later ones two levels.  Levels
    enqueued past the end are no-ops.  The batch schedule depends only on numbers every rank agrees on, so all ranks
    issue the same collectives.  `engine` needs reset/push/merge/status/labels."""

    def __init__(self, engine, rank, world, comm_device):
        self.e, self.rank, self.world, self.comm_device = engine, rank, world, torch.device(comm_device)
        self.levels_hint = 8
        self._gathered = None
        self._glists = None
        self._bufs = None
        self.sparse_levels = self.dense_levels = 0
        mode = os.environ.get("MGX_DIST_EXCHANGE", "auto")
        if mode not in ("gather", "reduce"):
            mode = "reduce" if world >= 4 else "gather"
        self.exchange = mode
        # The per-level loop inside the library over a communicator of its own (mgx_dbfs2_run) whenever the engine is
        # the HIP one and the collectives run on the GPUs (RCCL); the Python loop below serves the gloo tests, the
        # one-GPU pre-flights and MGX_DIST_NATIVE=0.
        self.native, self.comm, self.native_error = False, None, None
        if hasattr(engine, "run_native") and self.comm_device.type == "cuda" and os.environ.get("MGX_DIST_NATIVE", "1") != "0":
            ok = 1
            if world > 1 or os.environ.get("MGX_DIST_FORCE_COLLECTIVES") == "1":
                try:
                    self.comm = NativeComm(engine.ctx, rank, world, self.comm_device)
                except Exception as ex:                      # (librccl not loadable, symbol missing, init refused ...)
                    self.comm, ok, self.native_error = None, 0, repr(ex)
                if world > 1:
                    # every rank must take the same path: the Python loop below issues torch.distributed collectives, the
                    # native one RCCL calls of its own -- one rank without a communicator sends everybody to the Python loop
                    flag = torch.tensor([ok], dtype=torch.int32, device=self.comm_device)
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    ok = int(flag.item())
                    if not ok and self.comm is not None:
                        self.comm.close()
                        self.comm = None
            self.native = bool(ok)

    def _gather_lists(self, mine):
        """-> (lists, nlists): every rank's id list of the level (what apply_lists takes)"""
        W = self.world
        if W == 1 and os.environ.get("MGX_DIST_FORCE_COLLECTIVES") != "1":
            return mine, 1
        src = mine if mine.device == self.comm_device else mine.to(self.comm_device)
        if self._glists is None or self._glists.numel() != W * src.numel() or self._glists.device != self.comm_device:
            self._glists = torch.empty(W * src.numel(), dtype=src.dtype, device=self.comm_device)
        dist.all_gather_into_tensor(self._glists, src.contiguous())
        out = self._glists
        return (out if out.device == mine.device else out.to(mine.device)), W

    def _exchange(self, new):
        """-> (maps, nmaps): what merge() takes"""
        W = self.world
        if W == 1 and os.environ.get("MGX_DIST_FORCE_COLLECTIVES") != "1":    # (pre-flight: a group of one through RCCL)
            return new, 1
        mine = new if new.device == self.comm_device else new.to(self.comm_device)
        if self.exchange == "gather":
            if self._gathered is None or self._gathered.numel() != W * mine.numel() or self._gathered.device != self.comm_device:
                self._gathered = torch.empty(W * mine.numel(), dtype=mine.dtype, device=self.comm_device)
            dist.all_gather_into_tensor(self._gathered, mine.contiguous())
            out = self._gathered
            return (out if out.device == new.device else out.to(new.device)), W
        L = mine.numel()
        S = (L + W - 1) // W                      # words per slice (engines pad their maps: L % W == 0 then)
        if self._bufs is None or self._bufs[0].numel() != S * W or self._bufs[0].device != self.comm_device:
            self._bufs = tuple(torch.zeros(S * W, dtype=mine.dtype, device=self.comm_device) for _ in range(3))
        send, recv, full = self._bufs
        if L == S * W:
            send = mine.contiguous()
        else:
            send[:L].copy_(mine)
        dist.all_to_all_single(recv, send)        # recv slice r = rank r's bits of MY slice
        parts = recv.view(W, S)
        acc = parts[0]
        if hasattr(self.e, "or_maps") and recv.is_cuda and S % 4 == 0:
            self.e.or_maps(recv, W, acc)          # one kernel (out aliases map 0: every word is read before it is written)
        else:
            for r in range(1, W):
                torch.bitwise_or(acc, parts[r], out=acc)
        dist.all_gather_into_tensor(full, acc)    # every rank's merged slice: the OR of all maps
        out = full[:L] if L != S * W else full
        return (out if out.device == new.device else out.to(new.device)), 1

    def run(self, src):
        e = self.e
        if self.native:
            st = e.run_native(src, self.comm, self.exchange)
            self.levels = st["levels"]
            return {"levels": st["levels"], "edges_local": st["edges_local"]}
        e.reset(src)
        level = 0
        if getattr(e, "list", None) is not None:
            # one level per round: the id lists first; the bitmaps only when some rank's discoveries did not fit its list
            # (every rank reads the same headers, so all take the same branch); a level whose lists are all empty ends it
            self.sparse_levels = self.dense_levels = self.max_sparse_total = 0
            while True:
                new = e.push(level)
                overflow, total = e.apply_lists(level, *self._gather_lists(e.list))
                level += 1
                if total == 0:
                    break
                if overflow:
                    e.merge(level - 1, *self._exchange(new))
                    self.dense_levels += 1
                else:
                    self.sparse_levels += 1
                    self.max_sparse_total = max(self.max_sparse_total, int(total))
            st = e.status(level)
            self.levels = st["levels"]
            return {"levels": st["levels"], "edges_local": st["edges_local"]}
        batch = self.levels_hint
        while True:
            for _ in range(batch):
                new = e.push(level)
                e.merge(level, *self._exchange(new))
                level += 1
            st = e.status(level)
            if st["over"]:
                break
            batch = 2
        self.levels = st["levels"]
        self.levels_hint = max(1, st["levels"] + 1)      # + the level that finds nothing
        return {"levels": st["levels"], "edges_local": st["edges_local"]}

    def gather_labels(self):
        """Global label array in (hub-first) global ids on every rank (validation only)."""
        loc = torch.from_numpy(self.e.labels())
        n, W = self.e.n_global, self.world
        if W == 1:
            return loc.numpy()
        cap = (n + W - 1) // W
        pad = torch.full((cap,), -2, dtype=torch.int32)
        pad[: loc.numel()] = loc
        out = [torch.empty(cap, dtype=torch.int32, device=self.comm_device) for _ in range(W)]
        dist.all_gather(out, pad.to(self.comm_device))
        full = torch.stack([o.cpu() for o in out], dim=1).reshape(-1)[:n]   # vertex v = i*W + r
        return full.numpy()


def cyclic_shard_from_csr(ro, ci, ranks, rank):
    """Host (numpy) version of the generation-2 layout for a CSR that fits in memory: returns
    (row_offsets_local, col_indices_global_new, new_of_old, old_of_new).  Used by the tests."""
    ro = np.asarray(ro, dtype=np.int64)
    n = len(ro) - 1
    deg = np.diff(ro)
    old_of_new = np.argsort(-deg, kind="stable")
    new_of_old = np.empty(n, dtype=np.int64)
    new_of_old[old_of_new] = np.arange(n)
    mine_new = np.arange(rank, n, ranks)
    mine_old = old_of_new[mine_new]
    d = deg[mine_old]
    ro_l = np.zeros(len(mine_old) + 1, dtype=np.int64)
    np.cumsum(d, out=ro_l[1:])
    ci_l = np.empty(int(ro_l[-1]), dtype=np.int32)
    ci = np.asarray(ci)
    for i, v in enumerate(mine_old):
        ci_l[ro_l[i]:ro_l[i + 1]] = np.sort(new_of_old[ci[ro[v]:ro[v + 1]]])
    return ro_l.astype(np.int32), ci_l, new_of_old.astype(np.int32), old_of_new.astype(np.int32)


def rmat_cyclic_shard(ctx, scale, edgefactor, seed, ranks, rank, device):
    """Local CSR of the symmetrised R-MAT graph under the generation-2 layout: vertices renumbered
    hub-first by GLOBAL degree (every rank derives the same permutation from the same pair stream),
    vertex v owned by rank v % ranks, local row v // ranks, neighbour ids global.  Built inside the library
    (mgx_dbfs2_shard_plan / _fill: mgx_layout.hip); this function only allocates the result tensors.
    Returns (row_offsets_local, col_indices, new_of_old, old_of_new, degree_of_new) as device tensors."""
    n = 1 << scale
    plan, nl, ml = C.c_void_p(), C.c_int(), C.c_int64()
    check(lib.mgx_dbfs2_shard_plan(ctx._h, int(scale), int(edgefactor), C.c_uint64(seed), int(ranks), int(rank), C.byref(plan),
                                   C.byref(nl), C.byref(ml)))
    try:
        ro = torch.empty(nl.value + 1, dtype=torch.int32, device=device)
        col = torch.empty(max(ml.value, 1), dtype=torch.int32, device=device)[: ml.value]
        new_of_old = torch.empty(n, dtype=torch.int32, device=device)
        old_of_new = torch.empty(n, dtype=torch.int32, device=device)
        deg_new = torch.empty(n, dtype=torch.int32, device=device)
        torch.cuda.synchronize(device)
        check(lib.mgx_dbfs2_shard_fill(ctx._h, plan, C.c_void_p(ro.data_ptr()), C.c_void_p(col.data_ptr() if ml.value else 0),
                                       C.c_void_p(new_of_old.data_ptr()), C.c_void_p(old_of_new.data_ptr()), C.c_void_p(deg_new.data_ptr())))
    finally:
        lib.mgx_dbfs2_shard_free(plan)
    return ro, col, new_of_old, old_of_new, deg_new.to(torch.int64)


def rmat_cyclic_shard_torch(ctx, scale, edgefactor, seed, ranks, rank, device, pairs_per_chunk=1 << 26):
    """the same with torch device ops (the first implementation): kept as the cross-check of the library's builder
    (tests/test_dist.py)"""
    n = 1 << scale
    total = edgefactor * n
    deg = torch.zeros(n, dtype=torch.int64, device=device)

    def chunks():
        for first in range(0, total, pairs_per_chunk):
            cnt = min(pairs_per_chunk, total - first)
            s = torch.empty(cnt, dtype=torch.int32, device=device)
            d = torch.empty(cnt, dtype=torch.int32, device=device)
            torch.cuda.synchronize(device)
            api.rmat_edges(ctx, scale, first, cnt, seed, True, s, d, None)
            ctx.synchronize()
            yield s, d

    for s, d in chunks():                       # pass 1: global degrees (row = dst, and row = src for the swapped copy)
        deg += torch.bincount(d.to(torch.int64), minlength=n)
        deg += torch.bincount(s.to(torch.int64), minlength=n)
    old_of_new = torch.sort(deg, descending=True, stable=True).indices
    new_of_old = torch.empty_like(old_of_new)
    new_of_old[old_of_new] = torch.arange(n, device=device)
    keys = []
    for s, d in chunks():                       # pass 2: keep the entries whose (new) row this rank owns
        ns, nd = new_of_old[s.to(torch.int64)], new_of_old[d.to(torch.int64)]
        for row, nbr in ((nd, ns), (ns, nd)):
            m = (row % ranks) == rank
            keys.append(((row[m] // ranks) << 32) | nbr[m])
    key = torch.cat(keys)
    del keys
    key, _ = torch.sort(key)
    col = (key & 0xFFFFFFFF).to(torch.int32)
    lrow = key >> 32
    del key
    nl = local_rows(n, ranks, rank)
    counts = torch.bincount(lrow, minlength=nl)
    ro = torch.zeros(nl + 1, dtype=torch.int64, device=device)
    torch.cumsum(counts, 0, out=ro[1:])
    return ro.to(torch.int32), col, new_of_old.to(torch.int32), old_of_new.to(torch.int32), deg[old_of_new]


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
