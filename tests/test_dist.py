"""Multi-process tests of the vertex-range partitioned BFS (mini_amd/dist_bfs.py).
CPU (not gpu): world_size 2 and 3 over gloo with the numpy rank engine -- exercises partitioning,
the all_to_all_v exchange, the own-bin shortcut and the all_reduce termination.
GPU (-m gpu): the same driver with the HIP rank engine, 2 ranks sharing cuda:0, exchange staged
through gloo (the box has one GPU; RCCL itself only comes into play in bench.py --gpus N)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard(ro, ci, lo, hi):
    return (ro[lo:hi + 1] - ro[lo]).astype(np.int32), ci[ro[lo]:ro[hi]].astype(np.int32)


def _worker(rank, world, port, use_gpu, scale, seed, sources, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mini_amd.dist_bfs import DistBfs, range_of
    from tests.oracle_binding import Oracle
    orc = Oracle()
    n, ro, ci, _ = orc.rmat_csr(scale, 16, seed)
    lo, hi = range_of(n, world, rank)
    ro_l, ci_l = _shard(ro, ci, lo, hi)
    if use_gpu:
        import mini_amd
        from mini_amd.dist_bfs import HipRankEngine
        torch.cuda.set_device(0)
        ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
        eng = HipRankEngine(ctx, n, world, rank, torch.from_numpy(ro_l).cuda(), torch.from_numpy(ci_l).cuda())
    else:
        from tests.dist_cpu_engine import NumpyRankEngine
        eng = NumpyRankEngine(n, world, rank, ro_l, ci_l)
    bfs = DistBfs(eng, rank, world, "cpu")
    ok = True
    deg = np.diff(ro)
    for src in sources:
        st = bfs.run(src)
        got = bfs.gather_labels()
        want = orc.bfs_cpu(ro, ci, src)
        e = torch.tensor([st["edges_local"]], dtype=torch.int64)
        dist.all_reduce(e)
        ok = ok and np.array_equal(got, want) and int(e.item()) == int(deg[want >= 0].sum())
    if rank == 0:
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def _worker2(rank, world, port, use_gpu, scale, seed, sources, q):
    """generation 2: hub-first global ids, cyclic ownership, bitmap all-gather"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mini_amd.dist_bfs import DistBfs2, cyclic_shard_from_csr
    from tests.oracle_binding import Oracle
    orc = Oracle()
    n, ro, ci, _ = orc.rmat_csr(scale, 16, seed)
    ro_l, ci_l, new_of_old, old_of_new = cyclic_shard_from_csr(ro, ci, world, rank)
    if use_gpu:
        import mini_amd
        from mini_amd.dist_bfs import HipRankEngine2
        torch.cuda.set_device(0)
        ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
        eng = HipRankEngine2(ctx, n, world, rank, torch.from_numpy(ro_l).cuda(), torch.from_numpy(ci_l).cuda())
    else:
        from tests.dist_cpu_engine import NumpyRankEngine2
        eng = NumpyRankEngine2(n, world, rank, ro_l, ci_l, lists=os.environ.get("MGX_DIST_LISTS", "1") != "0")
    bfs = DistBfs2(eng, rank, world, "cpu")
    ok = True
    deg = np.diff(ro)
    sparse = dense = unit_levels = cold_levels = big_sparse = 0
    for src in sources:
        st = bfs.run(int(new_of_old[src]))
        sparse += bfs.sparse_levels
        big_sparse = max(big_sparse, getattr(bfs, "max_sparse_total", 0))
        dense += bfs.dense_levels
        if use_gpu:
            unit_levels += eng.dense_levels()
            cold_levels += eng.cold_levels()[0]
        got_new = bfs.gather_labels()
        got = np.empty(n, dtype=np.int32)
        got[old_of_new] = got_new
        want = orc.bfs_cpu(ro, ci, src)
        e = torch.tensor([st["edges_local"]], dtype=torch.int64)
        dist.all_reduce(e)
        good = np.array_equal(got, want) and int(e.item()) == int(deg[want >= 0].sum()) and st["levels"] == int(want.max()) + 1
        if not good and rank == 0:
            print("partitioned BFS differs: src %d labels equal %s (first difference at %s) edges %d want %d levels %d want %d; sparse %d dense %d"
                  % (src, np.array_equal(got, want), np.flatnonzero(got != want)[:5], int(e.item()), int(deg[want >= 0].sum()), st["levels"],
                     int(want.max()) + 1, bfs.sparse_levels, bfs.dense_levels), file=sys.stderr, flush=True)
        ok = ok and good
    if rank == 0 and use_gpu:
        print("partitioned BFS: sparse levels %d, bitmap levels %d, unit-block levels %d, cold-pass levels %d" % (sparse, dense, unit_levels, cold_levels),
              file=sys.stderr, flush=True)
    if os.environ.get("MGX_DIST_LISTS", "1") != "0":
        # id lists on the sparse levels, bitmaps where a rank's discoveries do not fit its list (252 ids on these graphs):
        # every run has sparse levels; the hub's big level overflows from scale 11 on
        ok = ok and sparse > 0 and (dense > 0 or scale < 11)
    else:
        ok = ok and sparse == 0
    if os.environ.get("MGX_TEST_EXPECT_BIG_SPARSE") is not None:
        # a sparse level whose ids need more than one workgroup round of k_d2_lists_apply (256 threads x 256 workgroups walk them)
        ok = ok and big_sparse > int(os.environ["MGX_TEST_EXPECT_BIG_SPARSE"])
    if use_gpu and os.environ.get("MGX_TEST_EXPECT_UNIT_LEVELS") is not None:
        # levels whose long rows were read from the rank's unit blocks (rank 0 answers for the job: it holds the first hub)
        ok = ok and ((unit_levels > 0) == (os.environ["MGX_TEST_EXPECT_UNIT_LEVELS"] == "1")) and ((eng.units > 0) == (os.environ.get("MGX_DIST_UNITS", "1") != "0"))
    if use_gpu and os.environ.get("MGX_TEST_EXPECT_COLD_LEVELS") is not None:
        ok = ok and ((cold_levels > 0) == (os.environ["MGX_TEST_EXPECT_COLD_LEVELS"] == "1")) and ((eng.cold_levels()[1] > 0) == (os.environ["MGX_TEST_EXPECT_COLD_LEVELS"] == "1"))
    if rank == 0:
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


def _run(world, use_gpu, scale, seed, worker=None, pick_sources=None):
    from tests.oracle_binding import Oracle
    n, ro, ci, _ = Oracle().rmat_csr(scale, 16, seed)
    deg = np.diff(ro)
    sources = [int(np.argmax(deg)), int(np.where(deg > 0)[0][-1]), int(np.where(deg == 0)[0][0])]
    if pick_sources is not None:
        sources = pick_sources(deg)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=worker or _worker, args=(r, world, port, use_gpu, scale, seed, sources, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, "rank exited with %s" % p.exitcode
    assert q.get(timeout=10) is True


@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_bfs_gloo_cpu(built, world):
    _run(world, False, 9, 9)


@pytest.mark.gpu
def test_partitioned_bfs_hip_engine_two_ranks_one_gpu(built):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _run(2, True, 12, 12)


@pytest.mark.parametrize("exchange", ["gather", "reduce"])
@pytest.mark.parametrize("world", [1, 2, 3])
def test_bitmap_exchange_bfs_gloo_cpu(built, world, exchange, monkeypatch):
    """exchange: one all_gather of the ranks' maps, or all_to_all of slices + OR + all_gather of the merged slices
    (DistBfs2; world 3 does not divide the bitmap: the slices are padded); bitmaps on every level (MGX_DIST_LISTS=0)"""
    monkeypatch.setenv("MGX_DIST_EXCHANGE", exchange)
    monkeypatch.setenv("MGX_DIST_LISTS", "0")
    _run(world, False, 9, 9, _worker2)


@pytest.mark.parametrize("world,scale,exchange", [(1, 9, "gather"), (2, 9, "gather"), (2, 12, "reduce"), (3, 12, "gather")])
def test_density_switched_exchange_bfs_gloo_cpu(built, world, scale, exchange, monkeypatch):
    """the default exchange: id lists all-gathered on sparse levels, the bitmap exchange only for a level on
    which some rank's discoveries overflow its list; every rank takes the same branch (they read the same headers) and the
    traversal ends on the level whose lists are all empty.  Labels == the oracle's, edges == m_t, levels == depth + 1."""
    monkeypatch.setenv("MGX_DIST_EXCHANGE", exchange)
    monkeypatch.setenv("MGX_DIST_LISTS", "1")
    _run(world, False, scale, scale, _worker2)


@pytest.mark.gpu
@pytest.mark.parametrize("world,scale,exchange", [(1, 12, "gather"), (2, 12, "gather"), (2, 16, "reduce"),
                                                  (3, 14, "gather"), (3, 14, "reduce")])
@pytest.mark.parametrize("lists", ["1", "0"])
def test_bitmap_exchange_bfs_hip_engine_ranks_share_one_gpu(built, world, scale, exchange, lists, monkeypatch):
    """the HIP rank engine, several ranks on the one GPU of the test box over gloo: id lists on sparse levels + bitmaps on
    dense ones (k_d2_newbits' list, k_d2_lists_apply), and bitmaps on every level"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    monkeypatch.setenv("MGX_DIST_EXCHANGE", exchange)
    monkeypatch.setenv("MGX_DIST_LISTS", lists)
    _run(world, True, scale, scale, _worker2)


@pytest.mark.gpu
@pytest.mark.parametrize("world,scale,lists,units,dense_div,expect", [
    (2, 14, "0", "1", "1000000", "1"),      # every level that has a long-row queue and a merged bitmap reads the unit blocks
    (3, 15, "0", "1", "1000000", "1"),
    (2, 15, "1", "1", None, None),          # the default rule (a quarter of the rank's units), lists on the sparse levels
    (1, 14, "0", "1", "1000000", "1"),
    (2, 14, "0", "0", None, "0")])          # no unit blocks: the queue walk on every level
def test_partitioned_ranks_read_big_levels_from_unit_blocks(built, world, scale, lists, units, dense_div, expect, monkeypatch):
    """mgx_dbfs2_build_units: the ranks' long rows as unit blocks with GLOBAL owners, the level's merged discoveries as the
    frontier bitmap (k_bfs_push_level -> bfs_dense_body); labels, edge counts and depths equal the oracle's whichever body
    a level takes (graphs inside the LDS prefix here: the cold-edge pass has its own test below)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    monkeypatch.setenv("MGX_DIST_EXCHANGE", "gather")
    monkeypatch.setenv("MGX_DIST_LISTS", lists)
    monkeypatch.setenv("MGX_DIST_UNITS", units)
    if dense_div is not None:
        monkeypatch.setenv("MGX_DIST_DENSE_DIV", dense_div)
    if expect is not None:
        monkeypatch.setenv("MGX_TEST_EXPECT_UNIT_LEVELS", expect)
    _run(world, True, scale, scale + 40, _worker2)


@pytest.mark.gpu
@pytest.mark.parametrize("world,scale,dense_div,cold,expect_cold,lists", [
    (2, 21, "1000000", "1", "1", "0"),       # three slices behind the LDS prefix: every level with a long-row queue runs the pass
    (3, 20, None, "1", None, "0"),           # the default rule (R-MAT 20: every vertex with edges is inside the prefix -- no lists are built)
    (3, 21, "1000000", "1", "1", "1"),       # with the density-switched exchange (id lists on the sparse levels)
    (2, 21, "1000000", "0", "0", "0")])      # MGX_DIST_COLD=0: cold entries are marked untested
def test_partitioned_ranks_cold_edge_pass(built, world, scale, dense_div, cold, expect_cold, lists, monkeypatch):
    """the ranks' cold-edge pass (bfs_fused_cold.hpp behind k_bfs_push_level; flush bitmaps ORed in by k_d2_newbits) on graphs
    whose id range outgrows the LDS prefix (R-MAT 20 / 21): labels, edge counts and depths equal the oracle's"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    monkeypatch.setenv("MGX_DIST_EXCHANGE", "gather")
    monkeypatch.setenv("MGX_DIST_LISTS", lists)
    monkeypatch.setenv("MGX_DIST_COLD", cold)
    if dense_div is not None:
        monkeypatch.setenv("MGX_DIST_DENSE_DIV", dense_div)
    if expect_cold is not None:
        monkeypatch.setenv("MGX_TEST_EXPECT_COLD_LEVELS", expect_cold)
    _run(world, True, scale, scale + 3, _worker2)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 2])
def test_sparse_level_with_more_ids_than_one_workgroup_round(built, world, monkeypatch):
    """on a one-rank run the gathered lists ARE the rank's own list; k_d2_lists_apply used to reset that
    list's count from workgroup 0 while later workgroups still had to read it -- a sparse level of more than 256 ids could lose
    vertices.  Sources of a few hundred neighbours on R-MAT 18 (list capacity 1024 / ranks): their first level is such a level."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    monkeypatch.setenv("MGX_DIST_EXCHANGE", "gather")
    monkeypatch.setenv("MGX_DIST_LISTS", "1")
    monkeypatch.setenv("MGX_TEST_EXPECT_BIG_SPARSE", "256")

    def pick(deg):
        lo, hi = 400, 900               # ~380-850 distinct neighbours: fits a rank's list (1024 / world ids), > 256 in all
        ids = np.where((deg >= lo) & (deg <= hi))[0]
        assert len(ids) >= 4
        return [int(v) for v in ids[:: max(1, len(ids) // 6)][:6]]
    for _ in range(2):                                   # (a scheduling race: more than one go)
        _run(world, True, 18, 58, _worker2, pick_sources=pick)


@pytest.mark.gpu
@pytest.mark.parametrize("world,scale", [(2, 14), (3, 13)])
def test_partitioned_ranks_list_based_queue_build(built, world, scale, monkeypatch):
    """MGX_DIST_BUILD_LIST=1: the ranks' queue build through an LDS list (k_bfs_build<., false>, what a rank falls back to when
    its row offsets are not 16-byte aligned) instead of k_bfs_build2<., DIST>: the same labels, edge counts and depths"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    monkeypatch.setenv("MGX_DIST_EXCHANGE", "gather")
    monkeypatch.setenv("MGX_DIST_LISTS", "0")
    monkeypatch.setenv("MGX_DIST_BUILD_LIST", "1")
    _run(world, True, scale, scale + 90, _worker2)


@pytest.mark.gpu
def test_or_maps_kernel_matches_numpy(built):
    """mgx_dbfs2_or_maps (the reduce step of the slice exchange, used when the collectives run on the GPU)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mini_amd
    from mini_amd.dist_bfs import HipRankEngine2
    ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
    ro = torch.tensor([0, 1, 2], dtype=torch.int32).cuda()           # two vertices joined by an edge
    ci = torch.tensor([1, 0], dtype=torch.int32).cuda()
    eng = HipRankEngine2(ctx, 2, 1, 0, ro, ci)
    rng = np.random.default_rng(5)
    for nmaps, words in ((1, 4), (2, 8), (3, 1028), (8, 65536)):
        maps = rng.integers(-2**31, 2**31 - 1, size=nmaps * words, dtype=np.int64).astype(np.int32)
        want = np.bitwise_or.reduce(maps.reshape(nmaps, words), axis=0)
        d = torch.from_numpy(maps).cuda()
        out = torch.empty(words, dtype=torch.int32).cuda()
        eng.or_maps(d, nmaps, out)
        ctx.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)
        if words:
            eng.or_maps(d, nmaps, d[:words])                          # in place, as DistBfs2 uses it
            ctx.synchronize()
            assert np.array_equal(d[:words].cpu().numpy(), want)
    with pytest.raises(mini_amd.MgxError):
        eng.or_maps(torch.zeros(6, dtype=torch.int32).cuda(), 2, torch.zeros(3, dtype=torch.int32).cuda())   # words % 4
    eng.close()


def _worker_sssp(rank, world, port, use_gpu, scale, seed, sources, q):
    """partitioned SSSP: range partition, (vertex, distance) pairs with min-combining before send"""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mini_amd.dist_bfs import range_of
    from mini_amd.dist_sssp import DistSssp
    from tests.oracle_binding import Oracle
    orc = Oracle()
    n, ro, ci, w = orc.rmat_csr(scale, 16, seed)
    lo, hi = range_of(n, world, rank)
    ro_l, ci_l = _shard(ro, ci, lo, hi)
    w_l = w[ro[lo]:ro[hi]].astype(np.float32)
    if use_gpu:
        import mini_amd
        from mini_amd.dist_sssp import HipSsspRankEngine
        torch.cuda.set_device(0)
        ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
        eng = HipSsspRankEngine(ctx, n, world, rank, torch.from_numpy(ro_l).cuda(), torch.from_numpy(ci_l).cuda(),
                                torch.from_numpy(w_l).cuda())
    else:
        from tests.dist_cpu_engine import NumpySsspRankEngine
        eng = NumpySsspRankEngine(n, world, rank, ro_l, ci_l, w_l)
    sssp = DistSssp(eng, rank, world, "cpu")
    ok = True
    for src in sources:
        st = sssp.run(src)
        got = sssp.gather_distances()
        want = orc.sssp_dijkstra_f32(ro, ci, w, src)          # (integer weights: every float32 path sum is exact)
        ok = ok and np.array_equal(got, want) and st["iterations"] >= 1
        # min-combining: a rank sends at most one pair per (target vertex, superstep)
        ok = ok and st["pairs_sent"] <= st["iterations"] * n
    if rank == 0:
        q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2, 3])
def test_partitioned_sssp_gloo_cpu(built, world):
    _run(world, False, 7, 21, _worker_sssp)


@pytest.mark.gpu
@pytest.mark.parametrize("world,scale", [(1, 12), (2, 12), (3, 14)])
def test_partitioned_sssp_hip_engine_ranks_share_one_gpu(built, world, scale):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    _run(world, True, scale, 30 + scale, _worker_sssp)


@pytest.mark.gpu
@pytest.mark.parametrize("collectives", ["1", "0"])
def test_partitioned_sssp_native_loop_one_rank(built, collectives, monkeypatch):
    """the superstep loop INSIDE the library (mgx_dsssp_run): with a one-rank RCCL communicator of the library's own the
    count / frontier-size all-gathers are issued from C++ (collectives 1), without one the same loop runs bare (0);
    distances bit-equal to the oracle's either way"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mini_amd
    from mini_amd.dist_sssp import DistSssp, HipSsspRankEngine
    from tests.oracle_binding import Oracle
    monkeypatch.setenv("MGX_DIST_FORCE_COLLECTIVES", collectives)
    orc = Oracle()
    n, ro, ci, w = orc.rmat_csr(13, 16, 77)
    ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
    eng = HipSsspRankEngine(ctx, n, 1, 0, torch.from_numpy(ro.astype(np.int32)).cuda(), torch.from_numpy(ci.astype(np.int32)).cuda(),
                            torch.from_numpy(w.astype(np.float32)).cuda())
    sssp = DistSssp(eng, 0, 1, "cuda")
    assert sssp.native, sssp.native_error
    assert (sssp.comm is not None) == (collectives == "1")
    for src in (0, 5, 4097):
        st = sssp.run(src)
        assert st["iterations"] >= 1 and st["pairs_sent"] == 0
        assert np.array_equal(sssp.gather_distances(), orc.sssp_dijkstra_f32(ro, ci, w, src))
    if sssp.comm is not None:
        sssp.comm.close()
    eng.close()


def test_bench_self_launch_starts_one_rank_per_gpu():
    """`python bench.py --gpus N` without a launcher starts N fresh ranks (torch.distributed.run on 127.0.0.1) before
    anything touches the GPU, instead of falling through to the 1-GPU body.  CPU: the ranks
    only report who they are (MGX_BENCH_LAUNCH_ONLY)."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["MGX_BENCH_LAUNCH_ONLY"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    import re
    seen = sorted(re.findall(r"launched rank (\d) of 3 \(--gpus 3\)", p.stdout + p.stderr))     # (ranks share the pipe: lines may interleave)
    assert seen == ["0", "1", "2"], p.stdout + p.stderr


def _bench_hang(extra_env, dist_timeout):
    import subprocess
    import sys
    import time as _t
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["MGX_BENCH_LAUNCH_ONLY"] = "hang"
    env.update(extra_env)
    t0 = _t.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--dist-timeout", str(dist_timeout)],
                       env=env, capture_output=True, text=True, timeout=300)
    return p, _t.time() - t0


def test_bench_rank_watchdog_ends_a_hung_rank():
    """first contact with 8 GPUs must not eat the caller's clock: a rank that never returns is
    ended by its OWN watchdog a little before --dist-timeout -- one line of reason, status 124 -- and the launcher, seeing
    a failed rank, takes the others down; the self-launcher passes the failure on"""
    p, took = _bench_hang({}, 40)
    assert p.returncode != 0 and took < 150, (p.returncode, took)
    assert "did not finish within" in p.stderr and "rank 1 of 2" in p.stderr, p.stderr[-3000:]


def test_bench_self_launch_ends_the_child_group_on_timeout():
    """... and when even the watchdogs do not fire in time (here: every rank hangs and the limit is shorter than the
    ranks' own margin), the self-launcher ends the whole child process group and exits 124 with its own one-line reason"""
    p, took = _bench_hang({"MGX_BENCH_HANG_ALL": "1", "MGX_BENCH_WATCHDOG_OFF": "1"}, 12)
    assert p.returncode == 124 and took < 90, (p.returncode, took, p.stderr[-2000:])
    assert "ending the child process group" in p.stderr


@pytest.mark.gpu
def test_bench_plain_command_runs_n_ranks(built):
    """the same on the GPU box, end to end: `python bench.py --gpus 2` (no torchrun) prints ONE line with n_gpus = 2 and
    rccl_ranks = 2 (two ranks on the one GPU over gloo: the pre-flight switches)"""
    import json
    import subprocess
    import sys
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env.update(MGX_BENCH_ALL_ON_GPU0="1", MGX_BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--scale", "14"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["rccl_ranks"] == 2 and j["parity_vs_oracle"] is True


def _bench_preflight(world, extra_args, env_extra):
    """bench.py's N > 1 body (bench_dist.bench_main) under torch.distributed.run with `world` ranks sharing the
    one GPU of the test box over gloo (RCCL refuses two ranks on one device): the launch contract, the partition, the
    exchange, the parity check and the JSON line -- everything of the multi-GPU bench except xGMI."""
    import json
    import socket
    import subprocess
    import sys
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MGX_BENCH_ALL_ON_GPU0="1", MGX_BENCH_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.update(env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "3", "--warmup", "1"] + extra_args
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
@pytest.mark.parametrize("world,extra,env_extra,scale_out,scaling", [
    (2, ["--scale", "14"], {}, 14, "strong"),
    (4, ["--scale", "13", "--scaling", "weak"], {}, 15, "weak"),
    (2, ["--scale", "14", "--no-cpu-baseline"], {"MGX_BENCH_TREE_CHECK": "1"}, 14, "strong"),
    (3, ["--scale", "12"], {"MGX_DIST_EXCHANGE": "reduce"}, 12, "strong")])
def test_bench_main_preflight_ranks_share_one_gpu(built, world, extra, env_extra, scale_out, scaling):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    j = _bench_preflight(world, extra, env_extra)
    assert j["n_gpus"] == world and j["scaling"] == scaling and j["config"]["scale"] == scale_out
    assert j["parity_vs_oracle"] is True and j["value"] > 0
    if "MGX_BENCH_TREE_CHECK" in env_extra:
        assert "BFS-tree" in j["parity_check"] and j["cpu_baseline"] is None
    else:
        assert "oracle" in j["parity_check"]
        if "--no-cpu-baseline" not in extra:
            assert j["cpu_baseline"]["value"] > 0 and j["cpu_baseline"]["cores"] == 1


@pytest.mark.gpu
def test_bench_main_carries_config5_next_to_the_strong_line(built):
    """at N = 8 the default line (strong RMAT-22) also carries `config5`: the same measurement on RMAT-26.  Pre-flight with
    two ranks on the one GPU over gloo and the branch forced (MGX_BENCH_CONFIG5=1), the second graph big enough (scale 24)
    to take the BFS-tree check of the sizes the oracle does not see"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    j = _bench_preflight(2, ["--scale", "15", "--no-cpu-baseline"], {"MGX_BENCH_CONFIG5": "1", "MGX_BENCH_CONFIG5_SCALE": "24"})
    assert j["n_gpus"] == 2 and j["config"]["scale"] == 15 and j["parity_vs_oracle"] is True
    c5 = j["config5"]
    assert c5["parity"] is True and "BFS-tree" in c5["parity_check"] and c5["value"] > 0 and c5["steps"] == 3
    assert "scale 24" in c5["workload"]
    # the line says which paths of the rank engine ran (round 4): on RMAT-24 over 2 ranks the cold-edge pass and a sparse level
    paths = c5["rank0_paths_last_traversal"]
    assert paths and "error" not in paths and paths["levels_appended_by_the_push"] >= 1 and paths["levels_from_unit_blocks"] >= 1, paths


@pytest.mark.gpu
@pytest.mark.parametrize("exchange,collectives", [("gather", "1"), ("reduce", "1"), ("gather", "0")])
def test_bench_main_native_rccl_loop_one_rank(built, exchange, collectives):
    """the per-level loop INSIDE the library over a communicator of its own (mgx_comm_*, mgx_dbfs2_run): bench.py's N > 1
    body under a one-rank RCCL group -- ncclCommInitRank, ncclAllGather and grouped ncclSend / ncclRecv issued from C++
    between the push and merge launches of a batch of levels (collectives 1), or the same loop without a communicator
    (0: what a single rank needs); parity against the oracle as in every bench line"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    j = _bench_preflight(1, ["--scale", "15"], {"MGX_BENCH_FORCE_DIST": "1", "MGX_BENCH_DIST_BACKEND": "nccl",
                                                "MGX_DIST_FORCE_COLLECTIVES": collectives, "MGX_DIST_EXCHANGE": exchange})
    assert j["n_gpus"] == 1 and j["parity_vs_oracle"] is True and j["value"] > 0
    assert j["config"]["native_loop"] is True
    # ... and beside the partitioned figure the replicas mode (every rank its own copy of the graph and its share of the sources)
    assert j["replicas"]["value"] > 0 and j["replicas"]["parity_vs_oracle"] is True


@pytest.mark.gpu
@pytest.mark.parametrize("scale,ranks", [(10, 1), (12, 3), (14, 8)])
def test_library_shard_builder_equals_the_torch_one(built, scale, ranks):
    """mgx_dbfs2_shard_plan / _fill (the rank's shard built inside the library: global degrees, hub-first permutation, the
    rank's rows as sorted keys) against the torch-op construction it replaces: the same row offsets, neighbour ids, id maps
    and degrees for every rank; and against the oracle's graph: relabelling the shard's rows back gives the CSR's rows"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mini_amd
    from mini_amd.dist_bfs import rmat_cyclic_shard, rmat_cyclic_shard_torch
    from tests.oracle_binding import Oracle
    ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
    dev = torch.device("cuda", 0)
    n, ro, ci, _ = Oracle().rmat_csr(scale, 16, 77 + scale)
    for rank in range(ranks):
        a = rmat_cyclic_shard(ctx, scale, 16, 77 + scale, ranks, rank, dev)
        b = rmat_cyclic_shard_torch(ctx, scale, 16, 77 + scale, ranks, rank, dev)
        for x, y, what in zip(a, b, ("row offsets", "neighbours", "new_of_old", "old_of_new", "degrees")):
            assert torch.equal(x.to(torch.int64), y.to(torch.int64)), (rank, what)
        ro_l, col, new_of_old, old_of_new = (t.cpu().numpy() for t in a[:4])
        assert ro_l[-1] == len(col)
        for i in (0, 1, len(ro_l) // 2, len(ro_l) - 2):
            if i < 0 or i >= len(ro_l) - 1:
                continue
            v_old = old_of_new[i * ranks + rank]
            want = np.sort(new_of_old[ci[ro[v_old]:ro[v_old + 1]]])
            assert np.array_equal(col[ro_l[i]:ro_l[i + 1]], want), (rank, i)
