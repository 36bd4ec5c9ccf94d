"""Synthetic R-MAT inputs, built on the device.

Edge pairs come from the counter-based HIP generator (mgx_rmat_edges, spec in
oracle/oracle.c:orc_rmat_edges).  Turning pairs into the reference's CSR is setup plumbing, not
the hot path, and uses torch device ops: it follows load_graph(_undir=true)
(gunrock/src/graph.hxx:129-172): a generated pair (u, v) is an MTX line "u v" -> CSR row v has
neighbour u; the swapped copy is appended; nothing is de-duplicated; rows sorted by neighbour id
(stable, so equal (row, neighbour) keep generation order, which fixes the order of weights).
"""
import torch

from . import api

A, B, C_, D = 0.57, 0.19, 0.19, 0.05


def _mix64_py(z):
    z = (z + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return z ^ (z >> 31)


def csr_from_pairs(src, dst, num_nodes, weight=None, undirected=True):
    """(src, dst[, weight]) int32 device tensors -> (row_offsets, col_indices, weights|None), all on device.

    Pair (u, v): row = v, neighbour = u.  undirected appends the swapped copies.
    """
    rows = torch.cat([dst, src]) if undirected else dst
    nbrs = torch.cat([src, dst]) if undirected else src
    w = None
    if weight is not None:
        w = torch.cat([weight, weight]) if undirected else weight
    key = (rows.to(torch.int64) << 32) | nbrs.to(torch.int64)
    del rows
    key, order = torch.sort(key, stable=True)
    col_indices = (key & 0xFFFFFFFF).to(torch.int32)
    row_sorted = (key >> 32)
    del key
    counts = torch.bincount(row_sorted, minlength=num_nodes)
    del row_sorted
    row_offsets = torch.zeros(num_nodes + 1, dtype=torch.int64, device=src.device)
    torch.cumsum(counts, 0, out=row_offsets[1:])
    weights = w[order] if w is not None else None
    del order
    return row_offsets.to(torch.int32), col_indices, weights


def rmat_csr(ctx, scale, edgefactor=16, seed=None, weighted=False, scramble=True, device=None, undirected=True,
             first_edge=0, num_pairs=None):
    """RMAT (a,b,c,d)=(.57,.19,.19,.05) CSR on `device` (default cuda:<ctx.device>).

    Returns dict(n, m, row_offsets, col_indices, weights, src, dst) of device tensors.
    seed defaults to `scale` (seed = 22 for config 2).
    """
    device = device or torch.device("cuda", ctx.device)
    seed = scale if seed is None else seed
    n = 1 << scale
    pairs = edgefactor * n if num_pairs is None else num_pairs
    src = torch.empty(pairs, dtype=torch.int32, device=device)
    dst = torch.empty(pairs, dtype=torch.int32, device=device)
    w = torch.empty(pairs, dtype=torch.float32, device=device) if weighted else None
    torch.cuda.synchronize(device)
    api.rmat_edges(ctx, scale, first_edge, pairs, seed, scramble, src, dst, w)
    ctx.synchronize()
    ro, ci, weights = csr_from_pairs(src, dst, n, w, undirected=undirected)
    torch.cuda.synchronize(device)
    return {"n": n, "m": int(ci.numel()), "row_offsets": ro, "col_indices": ci, "weights": weights}


def uniform_csr(ctx, scale, edgefactor=16, seed=None, device=None):
    """edgefactor * 2^scale pairs with both ends uniformly random (a seeded device generator), symmetrised like the R-MAT
    input (swapped copies appended, nothing de-duplicated): the same sizes as RMAT-<scale>, no hubs -- every row has about
    2 * edgefactor entries.  Returns the dict rmat_csr returns."""
    device = device or torch.device("cuda", ctx.device)
    seed = scale if seed is None else seed
    n = 1 << scale
    gen = torch.Generator(device=device)
    gen.manual_seed(0x5EED0000 + int(seed))
    pairs = edgefactor * n
    src = torch.randint(0, n, (pairs,), generator=gen, dtype=torch.int32, device=device)
    dst = torch.randint(0, n, (pairs,), generator=gen, dtype=torch.int32, device=device)
    ro, ci, _ = csr_from_pairs(src, dst, n, None, undirected=True)
    torch.cuda.synchronize(device)
    return {"n": n, "m": int(ci.numel()), "row_offsets": ro, "col_indices": ci, "weights": None}


def grid2d_csr(ctx, scale, device=None):
    """a 2^(scale // 2) x 2^(scale - scale // 2) grid, every vertex joined to its 4 neighbours (both directions stored):
    degree <= 4 everywhere and a diameter of rows + cols -- thousands of small levels.  Returns the dict rmat_csr returns."""
    device = device or torch.device("cuda", ctx.device)
    rows, cols = 1 << (scale // 2), 1 << (scale - scale // 2)
    n = rows * cols
    v = torch.arange(n, dtype=torch.int64, device=device)
    r, c = v // cols, v % cols
    right = v[c + 1 < cols]
    down = v[r + 1 < rows]
    src = torch.cat([right, down]).to(torch.int32)
    dst = torch.cat([right + 1, down + cols]).to(torch.int32)
    ro, ci, _ = csr_from_pairs(src, dst, n, None, undirected=True)
    torch.cuda.synchronize(device)
    return {"n": n, "m": int(ci.numel()), "row_offsets": ro, "col_indices": ci, "weights": None}


def degree_order(row_offsets, col_indices, weights=None):
    """Hub-first layout of a CSR (device tensors): vertex ids renumbered by descending degree (stable).

    Returns (layout_row_offsets, layout_col_indices, new_of_old, old_of_new), all int32 on the device --
    plus the weights in the layout's edge order when `weights` is given.
    Setup plumbing (untimed graph construction, like the CSR build itself); rows stay sorted by
    (new) neighbour id.
    """
    n = row_offsets.numel() - 1
    ro = row_offsets.to(torch.int64)
    deg = ro[1:] - ro[:-1]
    old_of_new = torch.sort(deg, descending=True, stable=True).indices
    new_of_old = torch.empty_like(old_of_new)
    new_of_old[old_of_new] = torch.arange(n, device=ro.device)
    rows_old = torch.repeat_interleave(torch.arange(n, device=ro.device), deg)
    key = (new_of_old[rows_old] << 32) | new_of_old[col_indices.to(torch.int64)]
    del rows_old
    key, order = torch.sort(key)
    lcol = (key & 0xFFFFFFFF).to(torch.int32)
    lw = weights[order] if weights is not None else None
    del key, order
    lro = torch.zeros(n + 1, dtype=torch.int64, device=ro.device)
    torch.cumsum(deg[old_of_new], 0, out=lro[1:])
    out = (lro.to(torch.int32), lcol, new_of_old.to(torch.int32), old_of_new.to(torch.int32))
    return out + (lw,) if weights is not None else out


def pick_sources(row_offsets_host, count, seed):
    """`count` vertices with degree > 0: splitmix64(seed + i) mod n, skipping isolated ones."""
    n = len(row_offsets_host) - 1
    out, i = [], 0
    while len(out) < count and i < 64 * count + 1024:
        v = _mix64_py(seed + i) % n
        i += 1
        if row_offsets_host[v + 1] > row_offsets_host[v]:
            out.append(int(v))
    return out
