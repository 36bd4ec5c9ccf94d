#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_dist.py tests/test_gpu_parity.py -q -m gpu --timeout 600 -x 2>&1 | tail -3
for G in 1 2 8; do
echo "-- scale 22 G=$G"; timeout 600 python tools/dist2_single.py 22 $G 2>&1 | tail -2
done
echo "-- scale 25 G=8"; timeout 900 python tools/dist2_single.py 25 8 2>&1 | tail -2
MGX_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 16 --warmup 2 2>&1 | tail -1 | cut -c1-220
