import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mini_amd
from tests.oracle_binding import Oracle
orc = Oracle()
ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 13
n, ro, ci, w = orc.rmat_csr(scale, 16, 160 + scale)
g = mini_amd.Graph.from_host(ctx, ro, ci) if hasattr(mini_amd.Graph, "from_host") else None
if g is None:
    g = mini_amd.Graph.from_device(ctx, n, len(ci), torch.from_numpy(ro).cuda(), torch.from_numpy(ci).cuda())
g.build_layout()
rng = np.random.default_rng(100 + scale)
ids = np.ascontiguousarray(np.sort(rng.permutation(n)[: n // 2]), dtype=np.int32)
vals = rng.integers(0, 8, size=n).astype(np.float32)
dv = torch.from_numpy(vals).cuda()
f = mini_amd.Frontier(ctx, n).load(ids)
red = torch.full((len(ids),), -1, dtype=torch.float32, device="cuda")
nz = mini_amd.segreduce(g, f, dv, 0.0, red, "f32_plus")
want, wnz = orc.neighbor_reduce_f32_plus(ro, ci, ids, vals, 0.0)
got = red.cpu().numpy()
bad = np.nonzero(got != want)[0]
deg = np.diff(ro)
print("n", n, "nf", len(ids), "nz", nz, wnz, "bad", len(bad))
for i in bad[:20]:
    print("pos", i, "vertex", ids[i], "deg", deg[ids[i]], "got", got[i], "want", want[i])
print("degrees of bad:", np.bincount(np.minimum(deg[ids[bad]], 70))[:71] if len(bad) else None)
print(g.nr_slices_info())
