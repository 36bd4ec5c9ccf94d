#!/usr/bin/env python3
"""bench.py -- BFS advance+filter MTEPS on synthetic R-MAT (BASELINE.json metric, config 2).

  python bench.py --gpus 1 --steps K --warmup W            (N>1: launched by torch.distributed.run)

A "step" is one whole BFS traversal (reset + every advance+filter level) from one seeded source
on the RMAT graph, inputs resident in HBM.  value = sum over steps of m_t (CSR entries of reached
vertices) / wall time of the K steps / 1e6, max over ranks.

Extra objects on the JSON line:
  roofline     dominant kernel k_bfs_push_level: algorithmic bytes (8 B/edge + 20 B/frontier vertex)
               per launch / average launch duration (HIP events on the launch stream), vs 8 TB/s.
  cpu_baseline the CPU oracle's restatement of bfs_problem_t::cpu (bfs_problem.hxx:52-72), one host
               thread, on a bounded sample of the same sources ("port": the reference itself
               cannot be built here).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=16)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scale", type=int, default=22)
    ap.add_argument("--edgefactor", type=int, default=16)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="budget of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--mode", choices=["push", "do"], default="push",
                    help="push = BASELINE config 2 (headline); do = direction-optimising (config 4)")
    ap.add_argument("--alpha", type=float, default=4.0, help="bottom-up switch: unvisited < frontier*alpha")
    ap.add_argument("--no-layout", action="store_true", help="keep generator vertex ids (no hub-first relabelling)")
    ap.add_argument("--pmc-json", default=os.path.join(ROOT, "profiles", "pmc_traffic.json"))
    return ap.parse_args()


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        args.gpus = world
    # Pre-flight switches for a one-GPU box (never a reported number): MGX_BENCH_ALL_ON_GPU0=1 puts every rank on
    # cuda:0, MGX_BENCH_DIST_BACKEND=gloo replaces RCCL (which refuses two ranks on one device)
    if os.environ.get("MGX_BENCH_ALL_ON_GPU0") == "1":
        local_rank = 0
    backend = os.environ.get("MGX_BENCH_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    # MGX_BENCH_FORCE_DIST=1 under a one-rank torchrun: run the N>1 code path (RCCL group of one) -- a
    # pre-flight for the multi-GPU bench on a one-GPU box, never a reported number
    force_dist = os.environ.get("MGX_BENCH_FORCE_DIST") == "1" and "RANK" in os.environ
    if world > 1 or force_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import __graft_entry__ as ge
    if rank == 0 and not os.path.exists(ge.LIB_OUT):
        ge.build()
    if world > 1:
        dist.barrier()
    import mini_amd
    from mini_amd import rmat

    if world > 1 or force_dist:
        from mini_amd import dist_bfs
        return dist_bfs.bench_main(args, rank, world, local_rank)

    stream = torch.cuda.current_stream()
    ctx = mini_amd.Context(local_rank, stream.cuda_stream)
    seed = args.scale if args.seed is None else args.seed
    t_build = time.time()
    g = rmat.rmat_csr(ctx, args.scale, args.edgefactor, seed=seed, weighted=False)
    graph = mini_amd.Graph.from_device(ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"])
    ro_host = g["row_offsets"].cpu().numpy()
    t_build = time.time() - t_build
    t_layout = time.time()
    if not args.no_layout:
        # hub-first layout (vertex ids by descending degree) for the LDS-resident hot bitmap; part of
        # graph construction like the CSR build, not of the timed traversal; labels stay in original ids
        graph.build_layout()          # mgx_graph_build_layout: device-side, inside the library
        torch.cuda.synchronize()
    t_layout = time.time() - t_layout
    sources = rmat.pick_sources(ro_host, args.steps + args.warmup, seed)
    bfs = mini_amd.BfsProblem(graph, sources[0])

    mode = mini_amd.MGX_BFS_DIRECTION_OPT if args.mode == "do" else mini_amd.MGX_BFS_PUSH
    for s in sources[:args.warmup]:
        bfs.run(s, mode, args.alpha)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    stats = []
    kernel_times = []
    t0 = time.perf_counter()
    ev0.record(stream)
    for s in sources[args.warmup:]:
        stats.append(bfs.run(s, mode, args.alpha))
    ev1.record(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    # Roofline pass: the SAME K sources again, now with HIP events around every launch of the two push kernels
    # (on the launch stream).  It is a second pass because every event record between two kernels leaves a
    # ~6 us gap on the stream (rocprofv3 kernel trace, profiles/), 3 events x ~8 levels per BFS: inside the timed
    # region they would cost ~10 % of `value`.
    bfs.set_kernel_timing(True)
    stats_timed = []
    for s in sources[args.warmup:]:
        stats_timed.append(bfs.run(s, mode, args.alpha))
        kernel_times.append(bfs.kernel_times())
    bfs.set_kernel_timing(False)

    m_t = sum(st["m_t"] for st in stats)
    reached = sum(st["reached"] for st in stats)
    launches = sum(st["kernel_launches"] for st in stats_timed)      # (batch events only exist in the second pass)
    kernel_ns = sum(st["kernel_ns"] for st in stats_timed)
    nf_total = sum(st["frontier_vertices"] for st in stats)   # vertices expanded (degree >= 1)
    # algorithmic bytes: top-down levels 8 B/edge expanded, bottom-up levels 4.125 B per inspected in-edge
    # (SURVEY 8d), 20 B per frontier vertex either way
    push_edges = sum(st["push_edges"] for st in stats)
    pull_edges = sum(st["pull_edges"] for st in stats)
    alg_bytes = 8.0 * push_edges + 4.125 * pull_edges + 20.0 * nf_total
    value = m_t / elapsed / 1e6

    # Dominant kernel: of the two push kernels (k_bfs_push_level_stream: rows of >= 64 edges read row-wise;
    # k_bfs_push_level_wave: shorter rows, searched per edge rank) the one with more device time in the timed
    # region.  achieved = algorithmic bytes of the edges / frontier vertices it processed / device time of
    # ALL its launches (HIP events around every launch on the launch stream; launches that find nothing to
    # do are included, as rocprofv3 --stats averages over them too).
    kt = {"stream": {"launches": 0, "ns": 0, "edges": 0, "vertices": 0}, "wave": {"launches": 0, "ns": 0, "edges": 0, "vertices": 0}}
    for k in kernel_times:
        for name in kt:
            for f in kt[name]:
                kt[name][f] += k[name][f]
    dom = "stream" if kt["stream"]["ns"] > kt["wave"]["ns"] else "wave"
    dom_launches, dom_ns = kt[dom]["launches"], kt[dom]["ns"]
    dom_edges, dom_vertices = kt[dom]["edges"], kt[dom]["vertices"]
    if dom_launches and dom_ns and dom_edges:
        kname = "k_bfs_push_level_" + dom
        dom_bytes = 8.0 * dom_edges + 20.0 * dom_vertices
        avg_launch_s = (dom_ns / 1e9) / dom_launches
        bytes_per_launch = dom_bytes / dom_launches
    else:                     # direction-optimising / other engines: all level kernels together
        kname = "bfs level kernels (all)"
        avg_launch_s = (kernel_ns / 1e9) / max(launches, 1)
        bytes_per_launch = alg_bytes / max(launches, 1)
    achieved = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
    traffic = None
    if os.path.exists(args.pmc_json):
        try:
            pj = json.load(open(args.pmc_json))
            if pj.get("scale") == args.scale and pj.get("kernel") == kname:
                traffic = pj.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 5),
                "traffic": traffic, "launches": dom_launches if dom_launches else launches,
                "timing": "HIP events around every launch of this kernel, second pass over the same %d sources "
                          "(events kept out of the timed region: each leaves a ~6 us gap on the stream; the timed "
                          "region launches this kernel's body and the short-row body as ONE grid per level, "
                          "k_bfs_push_level)" % len(stats),
                "avg_launch_us": round(avg_launch_s * 1e6, 3),
                "alg_bytes_per_launch": round(bytes_per_launch, 1),
                "share_of_edges": round(dom_edges / max(m_t, 1), 4) if dom_launches else 1.0,
                "all_level_kernels_alg_GBps": round(alg_bytes / max(kernel_ns / 1e9, 1e-12) / 1e9, 2),
                "whole_bfs_alg_GBps": round(alg_bytes / (dev_ms / 1e3) / 1e9, 2)}

    cpu = None
    parity = None
    if not args.no_cpu_baseline or not args.no_check:
        from tests.oracle_binding import Oracle
        orc = Oracle()
        ci_host = g["col_indices"].cpu().numpy()
        deg = np.diff(ro_host)
        cpu_edges, cpu_time, used = 0, 0.0, 0
        for s in sources[args.warmup:]:
            tc = time.perf_counter()
            want = orc.bfs_cpu(ro_host, ci_host, s)
            cpu_time += time.perf_counter() - tc
            cpu_edges += int(deg[want >= 0].sum())
            used += 1
            if used == 1 and not args.no_check:
                bfs.run(s, mode, args.alpha)
                parity = bool(np.array_equal(bfs.labels(), want))
            if cpu_time > args.cpu_seconds or args.no_cpu_baseline:
                break
        if not args.no_cpu_baseline:
            cpu = {"value": round(cpu_edges / cpu_time / 1e6, 2), "unit": "MTEPS", "cores": 1, "kind": "port",
                   "host_cpus": os.cpu_count(),
                   "sample": "oracle orc_bfs_cpu (restated bfs_problem_t::cpu) on %d of the %d timed sources, "
                             "same in-memory CSR, 1 thread, %.1f s" % (used, len(stats), cpu_time)}

    out = {"metric": "MTEPS (million traversed edges/sec) BFS advance+filter, RMAT-%d" % args.scale,
           "value": round(value, 2), "unit": "MTEPS", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(elapsed * 1e3 / max(args.steps, 1), 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "int32", "data": "synthetic",
           "config": {"workload": "BFS %s (fused LB advance + idempotent-visited filter) on RMAT scale %d ef %d, "
                                  "symmetrised, n=%d m=%d, %d seeded sources"
                                  % ("push" if args.mode == "push" else "direction-optimising alpha=%g" % args.alpha,
                                     args.scale, args.edgefactor, g["n"], g["m"], args.steps),
                      "scale": args.scale, "edgefactor": args.edgefactor, "seed": seed, "parallelism": "1 GPU",
                      "layout": "generator ids" if args.no_layout else "hub-first (degree-sorted) copy for the fused kernel"},
           "roofline": roofline, "cpu_baseline": cpu, "parity_vs_oracle": parity,
           "device_ms_per_step": round(dev_ms / max(args.steps, 1), 4),
           "avg_levels": round(sum(st["levels"] for st in stats) / max(len(stats), 1), 2),
           "avg_reached": reached // max(len(stats), 1), "graph_build_s": round(t_build, 2), "layout_build_s": round(t_layout, 2)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
