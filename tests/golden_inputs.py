"""Deterministic inputs of the larger golden cases (tests/golden/reference_goldens.json, "gen" entries).

The small cases are the reference's own fixture files (byte copies under tests/golden/).  The larger ones are R-MAT
graphs written as MatrixMarket text by the function below -- the same text every time (the JSON pins its sha256), so
the file itself need not be committed: tools/regen_goldens.py fed exactly this text to the reference's own load_graph /
cpu() validators, and the tests re-make it to run the oracle and the HIP path on the same input.

The graphs are SIMPLE (no self-loops, no parallel edges, one line per unordered pair): the reference's loader sorts
with a comparator that is not a strict weak order on equal keys (graph.hxx:139-157) -- harmless on its tiny fixtures,
undefined behaviour on thousands of duplicates.  Weights are the generator's integers in [0, 63]."""
import hashlib

import numpy as np

HEADER = "%%MatrixMarket matrix coordinate real general\n% mgx golden input: R-MAT scale {scale} ef {ef} seed {seed}, simple {kind}\n"


def rmat_simple_pairs(oracle, scale, edgefactor, seed, directed):
    """unique (row, col, weight) triples, 0-based, in generation order of their first occurrence"""
    n = 1 << scale
    s, d, w = oracle.rmat_edges(scale, 0, edgefactor * n, seed)
    s, d = s.astype(np.int64), d.astype(np.int64)
    keep = s != d
    s, d, w = s[keep], d[keep], w[keep]
    if not directed:
        lo, hi = np.minimum(s, d), np.maximum(s, d)
        s, d = lo, hi
    key = s * n + d
    _, first = np.unique(key, return_index=True)
    first.sort()
    return n, s[first].astype(np.int32), d[first].astype(np.int32), w[first]


def rmat_simple_mtx_text(oracle, scale, edgefactor, seed, directed=False):
    n, s, d, w = rmat_simple_pairs(oracle, scale, edgefactor, seed, directed)
    lines = [HEADER.format(scale=scale, ef=edgefactor, seed=seed, kind="directed" if directed else "undirected"),
             "%d %d %d\n" % (n, n, len(s))]
    lines += ["%d %d %d\n" % (a + 1, b + 1, int(x)) for a, b, x in zip(s.tolist(), d.tolist(), w.tolist())]
    return "".join(lines)


def sha(a):
    """sha256 of an array's little-endian bytes (int32 / float32) or of a str"""
    if isinstance(a, str):
        return hashlib.sha256(a.encode()).hexdigest()
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def case_path(case, oracle, tmp_dir, gold_dir):
    """the MatrixMarket file of a golden case: the committed fixture, or the re-made text (checked against its sha256)"""
    import os
    if "file" in case:
        return os.path.join(gold_dir, case["file"])
    g = case["gen"]
    text = rmat_simple_mtx_text(oracle, g["scale"], g["edgefactor"], g["seed"], g["directed"])
    assert sha(text) == case["mtx_sha256"], "golden input %s is not the text the reference was run on" % case["name"]
    path = os.path.join(str(tmp_dir), case["name"] + ".mtx")
    with open(path, "w") as f:
        f.write(text)
    return path


def matches(case, key, got, dtype):
    """does `got` equal the golden array `key` of the case (stored in full, or as sha256 of its bytes)?"""
    got = np.ascontiguousarray(got, dtype=dtype)
    if key in case:
        return got.tolist() == np.array(case[key], dtype=dtype).tolist()
    return sha(got) == case[key + "_sha256"]
