#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== DO alpha sweep"
for al in 4 64 1000; do
timeout 600 python bench.py --steps 16 --warmup 2 --no-cpu-baseline --mode do --alpha $al 2>&1 | tail -1 | cut -c1-200
done
echo "== dist2 single-GPU emulation"
for G in 1 2 8; do
echo "-- scale 22 G=$G"; timeout 600 python tools/dist2_single.py 22 $G 2>&1 | tail -2
done
echo "-- scale 25 G=8"; timeout 900 python tools/dist2_single.py 25 8 2>&1 | tail -2
echo "== dist bench path, 1 rank"
MGX_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 16 --warmup 2 2>&1 | tail -1 | cut -c1-300
echo "== sssp / pr"
timeout 600 python tools/sssp_bench.py 2>&1 | tail -3
timeout 600 python tools/pr_bench.py 2>&1 | tail -3
