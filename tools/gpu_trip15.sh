#!/bin/bash
mkdir -p gpurun_out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_dist.py -q -m gpu --timeout 600 -x 2>&1 | tail -3
MGX_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 16 --warmup 2 > gpurun_out/bench_dist1.log 2>&1
tail -3 gpurun_out/bench_dist1.log | cut -c1-900
for G in 8; do
echo "== dist2 scale 25 G=$G"
timeout 900 python tools/dist2_single.py 25 $G 2>&1 | tail -3
done
