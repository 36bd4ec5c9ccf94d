import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, mini_amd
from mini_amd import rmat
scale = int(sys.argv[1])
ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
g = rmat.rmat_csr(ctx, scale, 16, seed=scale, weighted=True)
graph = mini_amd.Graph.from_device(ctx, g["n"], g["m"], g["row_offsets"], g["col_indices"], g["weights"])
ro = g["row_offsets"].cpu().numpy()
srcs = rmat.pick_sources(ro, 2, scale)
sssp = mini_amd.SsspProblem(graph, srcs[0])
print(sssp.run(srcs[0]), flush=True)
