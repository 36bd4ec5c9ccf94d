"""CPU suite (-m "not gpu"): pins the oracle (oracle/oracle.c) to the reference's golden vectors
(tests/golden/reference_goldens.json) and checks the restatements against each other."""
import json
import os

import numpy as np
import pytest

from tests.golden_inputs import case_path, matches

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = json.load(open(os.path.join(GOLD, "reference_goldens.json")))["cases"]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_loader_and_cpu_validators_match_reference_goldens(oracle, case, tmp_path):
    """the oracle against what the reference's OWN load_graph / cpu() produced (tools/regen_goldens.sh): its fixtures with
    src 0, and simple R-MAT graphs of 1 K .. 16 K vertices from the busiest row"""
    n, ro, ci, w, srcs = oracle.load_mtx(case_path(case, oracle, tmp_path, GOLD), undir=case["undir"])
    src = case["src"]
    assert n == case["n"] and len(ci) == case["m"]
    assert matches(case, "offsets", ro, np.int32)
    assert matches(case, "indices", ci, np.int32)
    assert matches(case, "weights", w, np.float32)
    # csr.sources = row id per entry (graph.hxx:169)
    assert srcs.tolist() == np.repeat(np.arange(n), np.diff(ro)).tolist()
    assert matches(case, "bfs_labels", oracle.bfs_cpu(ro, ci, src), np.int32)          # bfs_problem.hxx:52-72
    preds, dist = oracle.sssp_cpu(ro, ci, w, src)                                      # sssp_problem.hxx:59-88
    assert matches(case, "sssp_preds", preds, np.int32)
    assert matches(case, "sssp_dist", dist, np.int32)
    if "kcore_largest" in case:                                                        # kcore_problem.hxx:54-105
        cores, largest = oracle.kcore_cpu(ro, ci)
        assert largest == case["kcore_largest"]
        assert matches(case, "kcore_num_cores", cores, np.int32)


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_enactor_restatements_agree_with_cpu_validators(oracle, case, tmp_path):
    n, ro, ci, w, _ = oracle.load_mtx(case_path(case, oracle, tmp_path, GOLD), undir=case["undir"])
    src = case["src"]
    want = oracle.bfs_cpu(ro, ci, src)
    assert matches(case, "bfs_labels", want, np.int32)
    # push only (alpha = 1/n, test_bfs.cu:30)
    rc, labels, stats = oracle.bfs_enact_pushpull(ro, ci, src, 1.0 / n)
    assert rc == 0 and labels.tolist() == want.tolist()
    if case["undir"] and len(ci) >= n:
        # the reference's pull phase walks the CSR copy (F8): right only on symmetric graphs
        for alpha in (0.05, 0.5, 1.0, 4.0):
            rc, labels, _ = oracle.bfs_enact_pushpull(ro, ci, src, alpha)
            assert rc == 0 and labels.tolist() == want.tolist(), alpha
    # SSSP: frontier Bellman-Ford fixed point == float Dijkstra == the reference's int distances
    dist, preds, st = oracle.sssp_enact(ro, ci, w, src, 1.5)
    dj = oracle.sssp_dijkstra_f32(ro, ci, w, src)
    assert np.array_equal(dist, dj)
    _, idist = oracle.sssp_cpu(ro, ci, w, src)
    reach = idist < np.iinfo(np.int32).max
    assert np.array_equal(dist[reach], idist[reach].astype(np.float32))
    assert np.all(dist[~reach] == np.finfo(np.float32).max)


def test_pull_phase_overflow_is_reported_not_exit(oracle):
    # n > m: the reference loads an n-int bitmap into an m-capacity frontier and exit(0)s (F14)
    ro = np.array([0, 1, 2, 2, 2, 2, 2, 2, 2], dtype=np.int32)
    ci = np.array([1, 0], dtype=np.int32)
    rc, _, _ = oracle.bfs_enact_pushpull(ro, ci, 0, 100.0)
    assert rc == -4


@pytest.mark.parametrize("scale,seed", [(8, 1), (10, 10), (12, 12), (14, 14)])
def test_rmat_restatements_consistent(oracle, scale, seed):
    n, ro, ci, w = oracle.rmat_csr(scale, 16, seed)
    assert len(ci) == 2 * 16 * n and ro[-1] == len(ci)
    # symmetric, rows sorted by neighbour
    rows = np.repeat(np.arange(n), np.diff(ro))
    assert np.all(np.diff(rows.astype(np.int64) * n + ci) >= 0)
    fwd = np.sort(rows.astype(np.int64) * n + ci)
    bwd = np.sort(ci.astype(np.int64) * n + rows)
    assert np.array_equal(fwd, bwd)
    assert w.min() >= 0 and w.max() <= 63 and np.all(w == np.floor(w))
    deg = np.diff(ro)
    src = int(np.argmax(deg))
    want = oracle.bfs_cpu(ro, ci, src)
    for alpha in (1.0 / n, 0.1, 2.0):
        rc, labels, _ = oracle.bfs_enact_pushpull(ro, ci, src, alpha)
        assert rc == 0 and np.array_equal(labels, want)
    dist, preds, _ = oracle.sssp_enact(ro, ci, w, src, 1.5)
    assert np.array_equal(dist, oracle.sssp_dijkstra_f32(ro, ci, w, src))
    # every finite vertex except src has a tight predecessor edge
    fin = np.where((dist < np.finfo(np.float32).max) & (np.arange(n) != src))[0]
    for v in fin[:200]:
        p = preds[v]
        es = np.arange(ro[p], ro[p + 1])
        assert np.any((ci[es] == v)), "pred is not a neighbour"


def test_scramble_is_a_bijection(oracle):
    for scale in (1, 5, 10, 16):
        ids = np.array([oracle.lib.orc_rmat_scramble(v, scale) for v in range(1 << min(scale, 12))])
        assert len(np.unique(ids)) == len(ids) and ids.max() < (1 << scale)


def test_scan_and_lbs_semantics(oracle):
    ro = np.array([0, 3, 3, 4, 4, 4, 9], dtype=np.int32)      # degrees 3 0 1 0 0 5
    fin = np.array([1, 0, 3, 5, 4, 2, 1], dtype=np.int32)     # ragged, with empty segments
    scanned, total = oracle.scan_degrees(ro, fin)
    assert scanned.tolist() == [0, 0, 3, 3, 8, 8, 9] and total == 9
    seg, rank = oracle.lbs(scanned, total)
    assert seg.tolist() == [1, 1, 1, 3, 3, 3, 3, 3, 5]
    assert rank.tolist() == [0, 1, 2, 0, 1, 2, 3, 4, 0]
    s0, t0 = oracle.scan_degrees(ro, np.zeros(0, dtype=np.int32))
    assert len(s0) == 0 and t0 == 0


def test_pr_restatement_first_iteration_is_a_pagerank_step(oracle):
    n, ro, ci, w, _ = oracle.load_mtx(os.path.join(GOLD, "pr_test.mtx"), undir=True)
    ranks, lens = oracle.pr_enact(ro, ci, 1)
    deg = np.diff(ro).astype(np.float32)
    want = np.float32(0.15) + np.float32(0.85) * (np.float32(0.15) * deg) / deg
    assert np.allclose(ranks, want, rtol=1e-6)
    assert len(lens) == 1


def _peel(ro, ci):
    """textbook peeling on the multigraph the CSR holds (every entry counts, a self-loop once): core numbers"""
    n = len(ro) - 1
    deg = np.diff(ro).astype(np.int64)
    core = np.zeros(n, dtype=np.int32)
    alive = deg > 0
    k = 1
    while alive.any():
        while True:
            rm = np.where(alive & (deg < k))[0]
            if len(rm) == 0:
                break
            core[rm] = k - 1
            alive[rm] = False
            for v in rm:
                np.subtract.at(deg, ci[ro[v]:ro[v + 1]], 1)
        k += 1
    return core


@pytest.mark.parametrize("case", [c for c in CASES if "kcore_largest" in c], ids=lambda c: c["name"])
def test_kcore_restatements_agree(oracle, case, tmp_path):
    """the enactor loop over serial operators (kcore_enactor.hxx:40-86) ends with the validator's core numbers
    (kcore_problem.hxx:54-105), and both are the textbook peeling of the loaded multigraph"""
    n, ro, ci, w, _ = oracle.load_mtx(case_path(case, oracle, tmp_path, GOLD), undir=True)
    cores, largest = oracle.kcore_cpu(ro, ci)
    ecores, elargest, st = oracle.kcore_enact(ro, ci)
    assert elargest == largest == case["kcore_largest"]
    assert np.array_equal(ecores, cores)
    assert st[0] == largest + 1 and st[3] == int((np.diff(ro) > 0).sum())   # every vertex with entries is removed once
    assert st[2] == len(ci)                                                # ... and expands its row once
    assert np.array_equal(cores, _peel(ro, ci))


def test_kcore_quirk_on_a_graph_without_entries(oracle):
    """cpu() answers 0 at k = 1 (nobody has degree >= 1); the enactor judges k = 1 by the count it started with (n) and
    never finds a pass that removes something: it runs to k = n and leaves largest_k_core at -1 (kcore_enactor.hxx:77-81)"""
    ro = np.zeros(6, dtype=np.int32)
    ci = np.zeros(0, dtype=np.int32)
    cores, largest = oracle.kcore_cpu(ro, ci)
    assert largest == 0 and not cores.any()
    ecores, elargest, st = oracle.kcore_enact(ro, ci)
    assert elargest == -1 and not ecores.any() and st[0] == 5
