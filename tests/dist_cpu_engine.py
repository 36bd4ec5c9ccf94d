"""A numpy stand-in for the per-rank device engine (mini_amd.dist_bfs.HipRankEngine), used ONLY by the
gloo CPU tests to drive the real superstep/exchange/termination logic of DistBfs without a GPU.
Same contract: expand -> per-owner bins of neighbour ids, each id at most once per traversal per rank."""
import numpy as np
import torch

from mini_amd.dist_bfs import chunk_of, range_of


class NumpyRankEngine:
    def __init__(self, n_global, ranks, rank, ro_local, ci_global):
        self.n_global, self.ranks, self.rank = n_global, ranks, rank
        self.lo, self.hi = range_of(n_global, ranks, rank)
        self.cap = chunk_of(n_global, ranks)
        self.ro = np.asarray(ro_local, dtype=np.int64)
        self.ci = np.asarray(ci_global, dtype=np.int64)
        self._bins = [np.zeros(0, dtype=np.int32)] * ranks

    def reset(self, src):
        self.lab = np.full(self.hi - self.lo, -1, dtype=np.int32)
        self.seen = np.zeros(self.n_global, dtype=bool)
        self.seen[src] = True
        self.front, self.next = [], []
        if self.lo <= src < self.hi:
            self.lab[src - self.lo] = 0
            self.front = [src - self.lo]

    def expand(self):
        nbrs = [self.ci[self.ro[v]:self.ro[v + 1]] for v in self.front]
        g = np.concatenate(nbrs) if nbrs else np.zeros(0, dtype=np.int64)
        edges = int(len(g))
        g = g[~self.seen[g]]
        _, first = np.unique(g, return_index=True)
        g = g[np.sort(first)]
        self.seen[g] = True
        owner = g // self.cap
        self._bins = [g[owner == r].astype(np.int32) for r in range(self.ranks)]
        return [len(b) for b in self._bins], edges

    def send_bin(self, r):
        return torch.from_numpy(self._bins[r].copy())

    def receive(self, ids, label):
        for gid in ids.tolist():
            v = gid - self.lo
            if self.lab[v] == -1:
                self.lab[v] = label
                self.next.append(v)

    def swap(self):
        self.front, self.next = self.next, []
        return len(self.front)

    def labels(self):
        return self.lab.copy()


class NumpyRankEngine2:
    """numpy stand-in for mini_amd.dist_bfs.HipRankEngine2 (generation 2: replicated visited bitmap,
    cyclic ownership, new-bit maps exchanged by all-gather).  Same reset/push/merge/status/labels contract,
    including "levels enqueued past the end are no-ops"."""

    def __init__(self, n_global, ranks, rank, ro_local, ci_global, lists=True):
        from mini_amd.dist_bfs import bitmap_words, list_words
        self.n_global, self.ranks, self.rank = n_global, ranks, rank
        self.n_local = (n_global - rank + ranks - 1) // ranks
        self.nwords = bitmap_words(n_global)
        # id lists of the sparse levels: [count, 0, 0, 0, ids ...] (count may exceed the capacity: "did not fit")
        self.list = torch.zeros(list_words(n_global, ranks), dtype=torch.int32) if lists else None
        self.ro = np.asarray(ro_local, dtype=np.int64)
        self.ci = np.asarray(ci_global, dtype=np.int64)

    @staticmethod
    def _bits_to_words(bits, nwords):
        pad = np.zeros(nwords * 32, dtype=np.uint8)
        pad[: len(bits)] = bits
        return np.packbits(pad.reshape(-1, 32)[:, ::-1], axis=1).view(">u4").astype(np.uint32).reshape(-1).view(np.int32)

    @staticmethod
    def _words_to_bits(words, n):
        w = np.asarray(words).view(np.uint32).astype(">u4")
        return np.unpackbits(w.view(np.uint8).reshape(-1, 4), axis=1)[:, ::-1].reshape(-1)[:n].astype(bool)

    def reset(self, src):
        self.lab = np.full(self.n_local, -1, dtype=np.int32)
        self.visited = np.zeros(self.n_global, dtype=bool)
        self.visited[src] = True
        self.front = []
        self.edges = 0
        self.over, self.levels, self.new_global = False, 0, -1
        if src % self.ranks == self.rank:
            i = src // self.ranks
            self.lab[i] = 0
            if self.ro[i + 1] > self.ro[i]:
                self.front = [i]

    def push(self, level):
        if level > 0 and self.new_global == 0 and not self.over:
            self.over, self.levels = True, level
        nbrs = [self.ci[self.ro[i]:self.ro[i + 1]] for i in self.front]
        g = np.concatenate(nbrs) if nbrs else np.zeros(0, dtype=np.int64)
        self.edges += len(g)
        new = np.zeros(self.n_global, dtype=bool)
        new[g[~self.visited[g]]] = True
        if self.list is not None:
            ids = np.nonzero(new)[0].astype(np.int32)
            cap = self.list.numel() - 4
            self.list.zero_()
            self.list[0] = len(ids)
            k = min(len(ids), cap)
            self.list[4:4 + k] = torch.from_numpy(ids[:k].copy())
        return torch.from_numpy(self._bits_to_words(new.astype(np.uint8), self.nwords).copy())

    def apply_lists(self, level, lists, nlists):
        L = lists.numpy().reshape(nlists, -1)
        cap = L.shape[1] - 4
        counts = L[:, 0].astype(np.int64)
        if (counts > cap).any():
            return True, int(np.minimum(counts, cap).sum())
        total = int(counts.sum())
        if total == 0:
            self.new_global = 0
            return False, 0
        ids = np.unique(np.concatenate([L[r, 4:4 + counts[r]] for r in range(nlists)]).astype(np.int64))
        ids = ids[~self.visited[ids]]
        self.visited[ids] = True
        own = ids[ids % self.ranks == self.rank] // self.ranks
        self.lab[own] = level + 1
        deg = self.ro[own + 1] - self.ro[own]
        self.front = own[deg > 0].tolist()
        self.new_global = len(ids)
        return False, total

    def merge(self, level, maps, nmaps):
        g = maps.numpy().reshape(nmaps, -1)[:, : self.nwords]
        merged = np.bitwise_or.reduce(g.view(np.uint32), axis=0)
        bits = self._words_to_bits(merged, self.n_global)
        self.visited |= bits
        mine = np.nonzero(bits[self.rank::self.ranks])[0]
        self.lab[mine] = level + 1
        deg = self.ro[mine + 1] - self.ro[mine]
        self.front = mine[deg > 0].tolist()
        self.new_global = int(bits.sum())

    def status(self, next_level):
        over = self.over or (next_level > 0 and self.new_global == 0)
        return {"over": over, "levels": self.levels if self.over else next_level, "edges_local": self.edges,
                "new_global": self.new_global}

    def labels(self):
        return self.lab.copy()


class NumpySsspRankEngine:
    """numpy stand-in for mini_amd.dist_sssp.HipSsspRankEngine (range partition, (vertex, distance) pairs with
    min-combining before send).  Same reset/expand/send_bin/receive/swap/distances contract."""
    device = torch.device("cpu")

    def __init__(self, n_global, ranks, rank, ro_local, ci_global, w_local):
        self.n_global, self.ranks, self.rank = n_global, ranks, rank
        self.lo, self.hi = range_of(n_global, ranks, rank)
        self.cap = chunk_of(n_global, ranks)
        self.ro = np.asarray(ro_local, dtype=np.int64)
        self.ci = np.asarray(ci_global, dtype=np.int64)
        self.w = np.asarray(w_local, dtype=np.float32)
        self._bins = [np.zeros(0, dtype=np.int64)] * ranks

    def reset(self, src):
        inf = np.float32(np.finfo(np.float32).max)
        self.dist = np.full(self.hi - self.lo, inf, dtype=np.float32)
        self.best = np.full(self.n_global, inf, dtype=np.float32)
        self.best[src] = 0.0
        self.front, self.next = [], set()
        if self.lo <= src < self.hi:
            self.dist[src - self.lo] = 0.0
            self.front = [src - self.lo]

    def expand(self):
        edges, touched = 0, set()
        for u in self.front:
            du = self.dist[u]
            for e in range(self.ro[u], self.ro[u + 1]):
                g, cand = int(self.ci[e]), np.float32(du + self.w[e])
                edges += 1
                if self.lo <= g < self.hi:
                    if cand < self.dist[g - self.lo]:
                        self.dist[g - self.lo] = cand
                        self.next.add(g - self.lo)
                elif cand < self.best[g]:
                    self.best[g] = cand
                    touched.add(g)
        self._bins = []
        for r in range(self.ranks):
            ids = sorted(g for g in touched if g // self.cap == r)
            bits = self.best[ids].view(np.uint32).astype(np.int64) if ids else np.zeros(0, dtype=np.int64)
            self._bins.append((np.asarray(ids, dtype=np.int64) << 32) | bits)
        return [len(b) for b in self._bins], edges

    def send_bin(self, r):
        return torch.from_numpy(self._bins[r].copy())

    def receive(self, pairs):
        p = pairs.numpy()
        for x in p.tolist():
            v = (x >> 32) - self.lo
            d = np.array([x & 0xFFFFFFFF], dtype=np.uint32).view(np.float32)[0]
            if d < self.dist[v]:
                self.dist[v] = d
                self.next.add(v)

    def swap(self):
        self.front, self.next = sorted(self.next), set()
        return len(self.front)

    def distances(self):
        return self.dist.copy()
