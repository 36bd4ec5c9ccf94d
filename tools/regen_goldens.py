#!/usr/bin/env python3
"""tools/regen_goldens.sh's second half: runs the golden driver (the reference's own load_graph / cpu() validators,
compiled by the shell script) on every golden input and writes tests/golden/reference_goldens.json.
usage: regen_goldens.py <driver> <workdir> <out.json>"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.golden_inputs import rmat_simple_mtx_text, sha  # noqa: E402
from tests.oracle_binding import Oracle  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")

# the reference's own fixtures (byte copies) with the loader arguments its drivers use (src = 0: test_bfs.cu:18)
SMALL = [("bfs_test_undir", "bfs_test.mtx", True), ("sssp_test_dir", "sssp_test.mtx", False),
         ("sssp_test_undir", "sssp_test.mtx", True), ("pr_test_undir", "pr_test.mtx", True),
         ("kcore_test_undir", "kcore_test.mtx", True), ("synthetic_dup_dir", "synthetic_dup.mtx", False)]
# larger inputs made by tests/golden_inputs.py (scale, edgefactor, seed, directed)
GEN = [("rmat10_undir", 10, 8, 10, False), ("rmat12_undir", 12, 8, 12, False), ("rmat14_undir", 14, 8, 14, False),
       ("rmat10_dir", 10, 8, 110, True)]


def run(driver, path, undir, src):
    out = subprocess.check_output([driver, path, "1" if undir else "0", str(src)], text=True)
    return json.loads(out)


def dist_from_preds(r):
    """the reference keeps its distances local (sssp_problem.hxx:68): recover them by walking its preds over its own CSR
    with its own int truncation of the weights; checks that the preds are a consistent tree"""
    n = r["n"]
    ro, ci, w, preds, src = r["offsets"], r["indices"], r["weights"], r["sssp_preds"], r["src"]
    INF = np.iinfo(np.int32).max
    dist = [None] * n
    dist[src] = 0

    def edge_w(p, v):
        ws = [int(w[e]) for e in range(ro[p], ro[p + 1]) if ci[e] == v]
        assert ws, "pred %d is not a neighbour of %d" % (p, v)
        return min(ws)
    for v in range(n):
        chain = []
        u = v
        while dist[u] is None and preds[u] >= 0:
            chain.append(u)
            u = preds[u]
        if dist[u] is None:
            dist[u] = INF                      # no pred, not the source: unreached
        for x in reversed(chain):
            p = preds[x]
            dist[x] = INF if dist[p] == INF else dist[p] + edge_w(p, x)
    return dist


def main():
    driver, work, out_path = sys.argv[1:4]
    orc = Oracle()
    cases = []
    for name, fname, undir in SMALL:
        r = run(driver, os.path.join(GOLD, fname), undir, 0)
        c = {"name": name, "file": fname, "undir": undir, "src": 0, "n": r["n"], "m": r["m"],
             "graph_t_undirected": r["graph_t_undirected"], "offsets": r["offsets"], "indices": r["indices"],
             "weights": r["weights"], "csc_equals_csr": r["csc_offsets"] == r["offsets"] and r["csc_indices"] == r["indices"],
             "bfs_labels": r["bfs_labels"], "sssp_preds": r["sssp_preds"], "sssp_dist": dist_from_preds(r)}
        if undir:                        # test_kcore.cu:24 loads undirected only
            c["kcore_num_cores"], c["kcore_largest"] = r["kcore_num_cores"], r["kcore_largest"]
        cases.append(c)
    for name, scale, ef, seed, directed in GEN:
        text = rmat_simple_mtx_text(orc, scale, ef, seed, directed)
        path = os.path.join(work, name + ".mtx")
        open(path, "w").write(text)
        n = 1 << scale
        # source: the vertex with the most CSR entries under the reference's own loader (ties: the smallest id)
        r0 = run(driver, path, not directed, 0)
        deg = np.diff(np.array(r0["offsets"]))
        src = int(np.argmax(deg))
        r = run(driver, path, not directed, src)
        lab = np.array(r["bfs_labels"], dtype=np.int32)
        c = {"name": name, "gen": {"scale": scale, "edgefactor": ef, "seed": seed, "directed": directed},
             "undir": not directed, "src": src, "n": r["n"], "m": r["m"], "graph_t_undirected": r["graph_t_undirected"],
             "mtx_sha256": sha(text),
             "offsets_sha256": sha(np.array(r["offsets"], dtype=np.int32)), "indices_sha256": sha(np.array(r["indices"], dtype=np.int32)),
             "weights_sha256": sha(np.array(r["weights"], dtype=np.float32)),
             "csc_equals_csr": r["csc_offsets"] == r["offsets"] and r["csc_indices"] == r["indices"],
             "bfs_labels_sha256": sha(lab), "sssp_preds_sha256": sha(np.array(r["sssp_preds"], dtype=np.int32)),
             "sssp_dist_sha256": sha(np.array(dist_from_preds(r), dtype=np.int32)),
             "bfs_reached": int((lab >= 0).sum()), "bfs_depth": int(lab.max()),
             "bfs_labels_head": r["bfs_labels"][:16], "sssp_preds_head": r["sssp_preds"][:16]}
        if not directed:
            c["kcore_num_cores_sha256"] = sha(np.array(r["kcore_num_cores"], dtype=np.int32))
            c["kcore_largest"] = r["kcore_largest"]
            c["kcore_num_cores_head"] = r["kcore_num_cores"][:16]
        cases.append(c)
    prov = ("Outputs of the reference's OWN load_graph / bfs_problem_t::cpu / sssp_problem_t::cpu / kcore_problem_t::cpu "
            "(gunrock/src/graph.hxx:96-223, bfs/bfs_problem.hxx:52-72, sssp/sssp_problem.hxx:59-88, kcore/kcore_problem.hxx:54-105; "
            "k-core on the undirected loads only, as test_kcore.cu:24 does), compiled from /root/reference where they lie by "
            "tools/regen_goldens.sh (g++; the three moderngpu includes are served by host stand-ins under tools/golden_ref/, "
            "own code) and run on (a) the reference's own test fixtures -- tests/golden/*.mtx are byte copies of "
            "gunrock/tests/{bfs,sssp,pr}/test.mtx and gunrock/tests/kcore/test_kcore.mtx; synthetic_dup.mtx is the "
            "survey's 3-line duplicate-edge probe -- with src = 0 (test_bfs.cu:18, test_sssp.cu:18), and (b) simple R-MAT "
            "graphs written as MatrixMarket text by tests/golden_inputs.py (pinned by mtx_sha256; arrays of these cases are "
            "stored as sha256 of their little-endian int32 / float32 bytes), src = the row with the most entries. "
            "sssp_dist is recovered from the reference's preds over the reference's CSR (its own distances are a local).")
    json.dump({"_provenance": prov, "_regenerate": "tools/regen_goldens.sh", "cases": cases}, open(out_path, "w"), indent=1)
    print("wrote %s: %d cases" % (out_path, len(cases)))


if __name__ == "__main__":
    main()
