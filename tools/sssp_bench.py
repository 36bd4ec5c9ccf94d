#!/usr/bin/env python3
"""SSSP (BASELINE config 3): RMAT weighted, operator-per-superstep path (mgx_sssp_enact) -- MTEPS = edge relaxations / s."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mini_amd
from mini_amd import rmat
ap = argparse.ArgumentParser(); ap.add_argument("--scale", type=int, default=22); ap.add_argument("--runs", type=int, default=4)
ap.add_argument("--queue-sizing", type=float, default=1.5); ap.add_argument("--check", type=int, default=1)
a = ap.parse_args()
ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
g = rmat.rmat_csr(ctx, a.scale, 16, seed=a.scale, weighted=True)
graph = mini_amd.Graph.from_device(ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"], g["weights"])
ro = g["row_offsets"].cpu().numpy()
srcs = rmat.pick_sources(ro, a.runs + 1, a.scale)
sssp = mini_amd.SsspProblem(graph, srcs[0])
sssp.enact(a.queue_sizing)
tot_t, tot_relax = 0.0, 0
for s in srcs[1:]:
    sssp.reset(s); ctx.synchronize()
    t0 = time.perf_counter(); st = sssp.enact(a.queue_sizing); ctx.synchronize(); dt = time.perf_counter() - t0
    tot_t += dt; tot_relax += st["relaxations"]
    print("src %d: iterations %d relaxations %d frontier_total %d  %.3f ms  %.1f MTEPS  alg %.1f GB/s" % (
        s, st["iterations"], st["relaxations"], st["frontier_total"], dt * 1e3, st["relaxations"] / dt / 1e6,
        (12.0 * st["relaxations"] + 24.0 * st["frontier_total"]) / dt / 1e9))
print("SSSP RMAT-%d operator path: %.1f MTEPS (relaxations/s) over %d sources, %.3f ms per source" % (a.scale, tot_relax / tot_t / 1e6, a.runs, tot_t / a.runs * 1e3))
op_dist = sssp.distances()
if os.environ.get("MGX_SSSP_LAYOUT", "1") != "0":
    graph.attach_layout(*rmat.degree_order(g["row_offsets"], g["col_indices"], g["weights"]))
sssp.run(srcs[0])
tot_t, tot_relax = 0.0, 0
for s in srcs[1:]:
    ctx.synchronize()
    t0 = time.perf_counter(); st = sssp.run(s); ctx.synchronize(); dt = time.perf_counter() - t0
    tot_t += dt; tot_relax += st["relaxations"]
    print("fused src %d: iterations %d relaxations %d frontier_total %d  %.3f ms  %.1f MTEPS" % (
        s, st["iterations"], st["relaxations"], st["frontier_total"], dt * 1e3, st["relaxations"] / dt / 1e6))
print("SSSP RMAT-%d fused loop: %.1f MTEPS (relaxations/s) over %d sources, %.3f ms per source; distances == operator path: %s" % (
    a.scale, tot_relax / tot_t / 1e6, a.runs, tot_t / a.runs * 1e3, bool(np.array_equal(sssp.distances(), op_dist))))
# near / far buckets (delta-stepping): fewer relaxations, more iterations
for delta in (4.0, 8.0, 16.0, 32.0, 64.0):
    sssp.run(srcs[0], delta=delta)
    tot_t, tot_relax, tot_it = 0.0, 0, 0
    for s in srcs[1:]:
        ctx.synchronize()
        t0 = time.perf_counter(); st = sssp.run(s, delta=delta); ctx.synchronize(); dt = time.perf_counter() - t0
        tot_t += dt; tot_relax += st["relaxations"]; tot_it += st["iterations"]
    print("SSSP RMAT-%d fused loop, near/far delta %g: %.3f ms per source, %.1f M relaxations, %.1f iterations per source; distances == operator path: %s" % (
        a.scale, delta, tot_t / a.runs * 1e3, tot_relax / a.runs / 1e6, tot_it / a.runs, bool(np.array_equal(sssp.distances(), op_dist))))
if a.check:
    from tests.oracle_binding import Oracle
    orc = Oracle(); ci = g["col_indices"].cpu().numpy(); w = g["weights"].cpu().numpy()
    t0 = time.perf_counter(); want, _, ost = orc.sssp_enact(ro, ci, w, srcs[-1], a.queue_sizing if a.queue_sizing > 4 else 8.0); dt = time.perf_counter() - t0
    print("oracle: %.2f s, %.1f MTEPS (1 thread); parity (bit-exact distances): %s" % (dt, ost[1] / dt / 1e6, bool(np.array_equal(sssp.distances(), want))))
