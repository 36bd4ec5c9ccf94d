"""Everything on a stream that is NOT the legacy default stream (torch.cuda.Stream() is created non-blocking: it is not
ordered against the NULL stream): the operator-per-superstep loops, the fused loops, the layout build and the read-backs
must all be ordered on the context's own stream (include/mgx/runtime.hpp: copies are hipMemcpyAsync on it)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture()
def stream_ctx(built, torch_mod):
    torch = torch_mod
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import mini_amd
    s = torch.cuda.Stream()
    ctx = mini_amd.Context(0, s.cuda_stream)
    yield ctx, s
    ctx.synchronize()
    ctx.close()


@pytest.mark.parametrize("scale", [10, 14])
def test_bfs_sssp_pr_on_a_non_default_stream(stream_ctx, oracle, scale):
    import mini_amd
    ctx, _ = stream_ctx
    n, ro, ci, w = oracle.rmat_csr(scale, 16, 77 + scale)
    w = np.floor(w).astype(np.float32)
    g = mini_amd.Graph.from_host(ctx, ro, ci, w)
    g.build_layout(weights=True)
    deg = np.diff(ro)
    srcs = [int(np.argmax(deg)), int(np.where(deg > 0)[0][7])]
    bfs = mini_amd.BfsProblem(g, srcs[0])
    sssp = mini_amd.SsspProblem(g, srcs[0])
    for src in srcs:
        want = oracle.bfs_cpu(ro, ci, src)
        for rep in range(3):                                   # back to back: nothing may lag behind on another stream
            bfs.run(src)
            assert np.array_equal(bfs.labels(), want)
        bfs.reset(src)
        bfs.enact_pushpull()
        assert np.array_equal(bfs.labels(), want)
        bfs.reset(src)
        bfs.enact_pushpull(4.0)
        assert np.array_equal(bfs.labels(), want)
        bfs.run(src, mode=mini_amd.MGX_BFS_DIRECTION_OPT, alpha=4.0)
        assert np.array_equal(bfs.labels(), want)
        dist, _, _ = oracle.sssp_enact(ro, ci, w, src, 8.0)
        sssp.run(src)
        assert np.array_equal(sssp.distances(), dist)
        sssp.reset(src)
        sssp.enact()
        assert np.array_equal(sssp.distances(), dist)
    pr = mini_amd.PrProblem(g, 3)
    lens = pr.enact()
    assert len(lens) >= 1 and np.all(np.isfinite(pr.ranks()))
    pr.close(); sssp.close(); bfs.close()
