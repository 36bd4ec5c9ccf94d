#!/bin/bash
cd $GRAFT_REPO_ROOT
ulimit -c 0
export MGX_BENCH_ALL_ON_GPU0=1 MGX_BENCH_DIST_BACKEND=gloo
for N in 2 4; do
echo "== preflight N=$N (gloo, all ranks on cuda:0, scale 18+log2 N)"
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2951$N bench.py --gpus $N --steps 8 --warmup 2 --scale 18 2>&1 | grep -v "amdgpu.ids\|OMP_NUM" | tail -3 | cut -c1-700
done
