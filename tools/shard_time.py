#!/usr/bin/env python3
"""time of one rank's shard construction (generation-2 layout): shard_time.py scale ranks rank"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mini_amd
from mini_amd.dist_bfs import rmat_cyclic_shard
scale, ranks, rank = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = mini_amd.Context(0, torch.cuda.current_stream().cuda_stream)
t0 = time.time()
ro, col, new_of_old, old_of_new, deg_new = rmat_cyclic_shard(ctx, scale, 16, scale, ranks, rank, torch.device("cuda", 0))
torch.cuda.synchronize()
print("scale %d rank %d/%d: %d rows %d edges in %.2f s, peak %.2f GB" % (scale, rank, ranks, ro.numel() - 1, col.numel(), time.time() - t0, torch.cuda.max_memory_allocated() / 2**30), flush=True)
