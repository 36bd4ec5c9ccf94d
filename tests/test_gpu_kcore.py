"""GPU suite (-m gpu): the repo's own k-core enactor (include/gunrock/kcore/, mgx_kcore_*) through the
C-ABI against the oracle's restatements of the reference's CPU validator (kcore_problem.hxx:54-105) and of its enactor
loop (kcore_enactor.hxx:40-86), on the reference's fixtures, the golden R-MAT inputs and R-MAT 10-16 with duplicate
edges and self-loops.  Integer work: bit-exact."""
import json
import os

import numpy as np
import pytest

from tests.golden_inputs import case_path, matches

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = [c for c in json.load(open(os.path.join(GOLD, "reference_goldens.json")))["cases"] if "kcore_largest" in c]


def _run(ctx, ro, ci):
    import mini_amd
    g = mini_amd.Graph.from_host(ctx, ro, ci, None)
    kc = mini_amd.KcoreProblem(g)
    largest, st = kc.enact()
    return kc, largest, st


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_kcore_matches_reference_goldens(gpu_ctx, oracle, case, tmp_path):
    """core numbers and largest k-core equal what the reference's own cpu() produced (tools/regen_goldens.sh)"""
    n, ro, ci, w, _ = oracle.load_mtx(case_path(case, oracle, tmp_path, GOLD), undir=True)
    kc, largest, st = _run(gpu_ctx, ro, ci)
    assert largest == case["kcore_largest"]
    assert matches(case, "kcore_num_cores", kc.num_cores(), np.int32)
    kc.close()


@pytest.mark.parametrize("scale,ef,seed", [(10, 16, 10), (12, 8, 12), (13, 16, 13), (14, 4, 14), (16, 16, 16)])
def test_kcore_rmat_parity(gpu_ctx, oracle, scale, ef, seed):
    """R-MAT as BASELINE's configs build it: symmetrised, duplicates and self-loops kept -- every entry is a degree"""
    n, ro, ci, w = oracle.rmat_csr(scale, ef, seed)
    want, wlargest = oracle.kcore_cpu(ro, ci)
    ecores, elargest, est = oracle.kcore_enact(ro, ci)
    assert wlargest == elargest and np.array_equal(want, ecores)
    kc, largest, st = _run(gpu_ctx, ro, ci)
    assert largest == wlargest
    assert np.array_equal(kc.num_cores(), want)
    # the same operator sequence as the serial restatement: k values, passes, expanded entries, removed vertices
    assert [st["rounds"], st["passes"], st["expanded"], st["removed"]] == est.tolist()
    assert st["expanded"] == len(ci) and st["removed"] == int((np.diff(ro) > 0).sum())
    assert np.all(kc.degrees() <= 0)                   # the run consumed the working degrees
    # a second run needs a reset, and gives the same answer
    kc.reset()
    assert np.array_equal(kc.degrees(), np.diff(ro)) and not kc.num_cores().any()
    largest2, st2 = kc.enact()
    assert largest2 == largest and st2 == st and np.array_equal(kc.num_cores(), want)
    kc.close()


def test_kcore_directed_and_ragged_inputs(gpu_ctx, oracle):
    """not symmetric, vertices without entries, a self-loop, parallel entries: the operators do what the serial
    restatement does (a directed input has no k-core meaning; the sequence is still defined)"""
    rng = np.random.default_rng(7)
    n = 500
    deg = rng.integers(0, 9, size=n)
    deg[rng.integers(0, n, size=60)] = 0
    ro = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    ci = rng.integers(0, n, size=int(ro[-1])).astype(np.int32)
    ci[ro[3]:ro[4]] = 3
    ecores, elargest, est = oracle.kcore_enact(ro, ci)
    kc, largest, st = _run(gpu_ctx, ro, ci)
    assert largest == elargest and np.array_equal(kc.num_cores(), ecores)
    assert [st["rounds"], st["passes"], st["expanded"], st["removed"]] == est.tolist()
    kc.close()


def test_kcore_graph_without_entries_keeps_the_upstream_quirk(gpu_ctx, oracle):
    ro = np.zeros(6, dtype=np.int32)
    ci = np.zeros(0, dtype=np.int32)
    _, elargest, est = oracle.kcore_enact(ro, ci)
    kc, largest, st = _run(gpu_ctx, ro, ci)
    assert largest == elargest == -1 and st["rounds"] == est[0] == 5 and not kc.num_cores().any()
    kc.close()
