#!/usr/bin/env python3
"""k-core decomposition (mgx_kcore_enact: the reference's peeling loop on filter / advance<has_output=false> / filter)
on R-MAT, timed, against the oracle's restatement of kcore_problem_t::cpu.  usage: kcore_bench.py [scale] [edgefactor]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mini_amd
from mini_amd import rmat
from tests.oracle_binding import Oracle
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 18
ef = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
g = rmat.rmat_csr(ctx, scale, ef, seed=scale)
graph = mini_amd.Graph.from_device(ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"])
kc = mini_amd.KcoreProblem(graph)
kc.enact(); kc.reset(); ctx.synchronize()
t0 = time.perf_counter(); largest, st = kc.enact(); ctx.synchronize(); dt = time.perf_counter() - t0
print("k-core RMAT-%d ef %d (n = %d, m = %d): largest k-core %d, %d k values, %d passes, %.1f ms (%.1f M entries/s, %.1f us per pass)"
      % (scale, ef, g["n"], g["m"], largest, st["rounds"], st["passes"], dt * 1e3, st["expanded"] / dt / 1e6, dt * 1e6 / max(st["passes"], 1)))
if scale <= 20:
    ro, ci = g["row_offsets"].cpu().numpy(), g["col_indices"].cpu().numpy()
    t0 = time.perf_counter(); want, wl = Oracle().kcore_cpu(ro, ci); dc = time.perf_counter() - t0
    print("oracle (restated kcore_problem_t::cpu, 1 thread): %.1f ms; core numbers equal: %s, largest equal: %s"
          % (dc * 1e3, bool(np.array_equal(kc.num_cores(), want)), wl == largest))
