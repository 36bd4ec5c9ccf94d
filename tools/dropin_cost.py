#!/usr/bin/env python3
"""What "drop-in" costs: one BFS traversal of the same R-MAT graph from the same sources
  (i)   the reference's UNCHANGED bfs_enactor_t + bfs_functor_t (a CAS per edge) on this repo's advance / filter operators
        (tests/dropin/_bin/ref_bench_bfs: warm, best of 3 per source),
  (ii)  the repo's restatement of the functor (reads the label before the CAS) on the same operators (mgx_bfs_enact_pushpull),
  (iii) the fused device-resident traversal behind the same data model (mgx_bfs_run),
each with its algorithmic rate against the 8 TB/s roof (8 B per traversed edge + 20 B per reached vertex).
   python tools/dropin_cost.py [scale] [sources]"""
import os, struct, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mini_amd
from mini_amd import rmat
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nsrc = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
g = rmat.rmat_csr(ctx, scale, 16, seed=scale)
ro, ci = g["row_offsets"].cpu().numpy(), g["col_indices"].cpu().numpy()
deg = np.diff(ro)
srcs = rmat.pick_sources(ro, nsrc, scale)
PEAK = 8000.0
def frac(m_t, reached, ms):
    return (8.0 * m_t + 20.0 * reached) / (ms * 1e-3) / 1e9 / PEAK
rows = []
# (i) the reference's own enactor and functor
exe = os.path.join(ROOT, "tests", "dropin", "_bin", "ref_bench_bfs")
ref_ms = {}
if os.path.exists(exe):
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "graph.bin")
        with open(path, "wb") as f:
            f.write(struct.pack("<iiq", g["n"], 0, g["m"]))
            ro.astype(np.int32).tofile(f)
            ci.astype(np.int32).tofile(f)
        r = subprocess.run([exe, path] + [str(s) for s in srcs], capture_output=True, text=True, timeout=1200)
        for ln in r.stdout.splitlines():
            if ln.startswith("RESULT"):
                p = ln.split()
                ref_ms[int(p[2])] = float(p[6])
        correct = r.stdout.count("Correct.")
        print("ref_bench_bfs: rc %d, %d of %d sources validated by the reference's own cpu()" % (r.returncode, correct, len(srcs)))
        if r.returncode != 0:
            print(r.stdout[-800:], r.stderr[-800:])
graph = mini_amd.Graph.from_device(ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"])
graph.build_layout()
bfs = mini_amd.BfsProblem(graph, srcs[0])
bfs.reset(srcs[0]); bfs.enact_pushpull(); bfs.run(srcs[0]); ctx.synchronize()
tot = {"i": [0.0, 0, 0], "ii": [0.0, 0, 0], "iii": [0.0, 0, 0]}
for s in srcs:
    best_op = 1e30
    for rep in range(3):
        bfs.reset(s); ctx.synchronize()
        t0 = time.perf_counter(); bfs.enact_pushpull(); ctx.synchronize(); best_op = min(best_op, (time.perf_counter() - t0) * 1e3)
    lab = bfs.labels()
    m_t, reached = int(deg[lab >= 0].sum()), int((lab >= 0).sum())
    best_f = 1e30
    for rep in range(3):
        ctx.synchronize(); t0 = time.perf_counter(); bfs.run(s); ctx.synchronize(); best_f = min(best_f, (time.perf_counter() - t0) * 1e3)
    assert np.array_equal(bfs.labels(), lab)
    for key, ms in (("i", ref_ms.get(s)), ("ii", best_op), ("iii", best_f)):
        if ms is not None:
            tot[key][0] += ms; tot[key][1] += m_t; tot[key][2] += reached
names = {"i": "(i)   reference's unchanged enactor + functor (CAS per edge) on our operators", "ii": "(ii)  repo's functor on the operator path (one operator call per superstep step)",
         "iii": "(iii) fused device-resident traversal (mgx_bfs_run, one call per source)"}
print("RMAT-%d ef 16, %d sources, best of 3 per source:" % (scale, len(srcs)))
for key in ("i", "ii", "iii"):
    ms, m_t, reached = tot[key]
    if ms > 0:
        print("  %-86s %8.3f ms per traversal  %7.1f GTEPS  %.4f of the 8 TB/s roof" % (names[key], ms / len(srcs), m_t / (ms * 1e-3) / 1e9, frac(m_t, reached, ms)))
