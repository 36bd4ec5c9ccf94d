#!/usr/bin/env python3
"""Neighbourhood-reduce / PR timing on RMAT (row f1 of SURVEY 8f): one pr iteration = segreduce over all edges + filter."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import mini_amd
from mini_amd import rmat
ap = argparse.ArgumentParser(); ap.add_argument("--scale", type=int, default=22); ap.add_argument("--iters", type=int, default=3)
a = ap.parse_args()
ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
g = rmat.rmat_csr(ctx, a.scale, 16, seed=a.scale)
graph = mini_amd.Graph.from_device(ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"])
f = mini_amd.Frontier(ctx, g["n"]).fill_iota(g["n"])
vals = torch.rand(g["n"], device="cuda")
red = torch.empty(g["n"], device="cuda")
mini_amd.segreduce(graph, f, vals, 0.0, red, "f32_plus"); ctx.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); nz = mini_amd.segreduce(graph, f, vals, 0.0, red, "f32_plus"); ctx.synchronize(); dt = time.perf_counter() - t0
    print("segreduce f32_plus all vertices: %d edges %.3f ms  %.1f GTEPS  alg %.1f GB/s" % (nz, dt * 1e3, nz / dt / 1e9, (8.0 * nz + 16.0 * g["n"]) / dt / 1e9))
want = torch.zeros(g["n"], device="cuda").index_add_(0, torch.repeat_interleave(torch.arange(g["n"], device="cuda"), (g["row_offsets"][1:] - g["row_offsets"][:-1]).long()), vals[g["col_indices"].long()])
print("max rel err vs torch index_add: %.3g" % float(((red - want).abs() / want.abs().clamp(min=1)).max()))
pr = mini_amd.PrProblem(graph, a.iters)
t0 = time.perf_counter(); lens = pr.enact(); ctx.synchronize(); dt = time.perf_counter() - t0
print("pr enact %d iterations: %.3f ms (%s)" % (len(lens), dt * 1e3, lens))
# The same operators on the hub-first relabelled copy of the graph (a plain CSR as far as they are concerned): the
# value gather then finds the hubs' values -- the targets of most edges -- next to each other in L2.
lro, lci, new_of_old, old_of_new = rmat.degree_order(g["row_offsets"], g["col_indices"])
lgraph = mini_amd.Graph.from_device(ctx, g["n"], g["m"], lro, lci)
lvals = vals[old_of_new.long()].contiguous()
lred = torch.empty(g["n"], device="cuda")
mini_amd.segreduce(lgraph, f, lvals, 0.0, lred, "f32_plus"); ctx.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); nz = mini_amd.segreduce(lgraph, f, lvals, 0.0, lred, "f32_plus"); ctx.synchronize(); dt = time.perf_counter() - t0
    print("hub-first copy: segreduce f32_plus all vertices: %d edges %.3f ms  %.1f GTEPS  alg %.1f GB/s" % (nz, dt * 1e3, nz / dt / 1e9, (8.0 * nz + 16.0 * g["n"]) / dt / 1e9))
print("hub-first copy: max rel err vs original order: %.3g" % float(((lred[new_of_old.long()] - want).abs() / want.abs().clamp(min=1)).max()))
lpr = mini_amd.PrProblem(lgraph, a.iters)
t0 = time.perf_counter(); lens = lpr.enact(); ctx.synchronize(); dt = time.perf_counter() - t0
print("hub-first copy: pr enact %d iterations: %.3f ms (%s)" % (len(lens), dt * 1e3, lens))
# ... and with the library's own layout on the graph (mgx_graph_build_layout): a full frontier takes mgx/nreduce.hpp -- since late round 5 the
# long rows by slice of their destinations -- inside the operator, results in generator ids
graph.build_layout()
mini_amd.segreduce(graph, f, vals, 0.0, red, "f32_plus"); ctx.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); nz = mini_amd.segreduce(graph, f, vals, 0.0, red, "f32_plus"); ctx.synchronize(); dt = time.perf_counter() - t0
    print("library layout: segreduce f32_plus all vertices: %d edges %.3f ms  %.1f GTEPS  alg %.1f GB/s" % (nz, dt * 1e3, nz / dt / 1e9, (8.0 * nz + 16.0 * g["n"]) / dt / 1e9))
print("library layout: max rel err vs torch index_add: %.3g; slices %s" % (float(((red - want).abs() / want.abs().clamp(min=1)).max()), graph.nr_slices_info()))
mini_amd.PrProblem(graph, a.iters).enact(); ctx.synchronize()        # (warm-up on a problem of its own)
ppr = mini_amd.PrProblem(graph, a.iters)
t0 = time.perf_counter(); lens = ppr.enact(); ctx.synchronize(); dt = time.perf_counter() - t0
print("library layout: pr enact %d iterations: %.3f ms (%s) -- PR's frontier is the vertices that have edges, a full frontier only on graphs without isolated vertices" % (len(lens), dt * 1e3, lens))
