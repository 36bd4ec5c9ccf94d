#!/usr/bin/env python3
"""Per-iteration sizes and times of the fused SSSP loop on the weighted R-MAT (device-side stamps: no host sync per iteration).
   python tools/sssp_iterations.py [--scale 22] [--runs 2]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mini_amd
from mini_amd import rmat
ap = argparse.ArgumentParser(); ap.add_argument("--scale", type=int, default=22); ap.add_argument("--runs", type=int, default=2)
a = ap.parse_args()
ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
g = rmat.rmat_csr(ctx, a.scale, 16, seed=a.scale, weighted=True)
graph = mini_amd.Graph.from_device(ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"], g["weights"])
graph.build_layout(weights=True)
srcs = rmat.pick_sources(g["row_offsets"].cpu().numpy(), a.runs + 1, a.scale)
sssp = mini_amd.SsspProblem(graph, srcs[0])
sssp.run(srcs[0])
for s in srcs[1:]:
    st = sssp.run(s)
    print("src %d: %s" % (s, st))
    for i, (nf, ne, ms) in enumerate(sssp.iteration_trace()):
        print("  it %2d  frontier %9d  edges %10d  %.3f ms  %.1f G relax/s" % (i, nf, ne, ms, ne / ms / 1e6 if ms > 0 else 0.0))
