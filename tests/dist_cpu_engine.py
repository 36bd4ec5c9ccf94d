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
