#!/bin/bash
ulimit -c 0
O=gpurun_out/r2d; rm -rf $O; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu --timeout 1200 --durations=12 > $O/pytest_gpu.log 2>&1
echo "pytest rc=$?"; tail -30 $O/pytest_gpu.log
timeout 600 python bench.py --steps 16 --warmup 2 > $O/bench.log 2>$O/bench.err
echo "bench rc=$?"; tail -1 $O/bench.log; tail -3 $O/bench.err
timeout 300 python bench.py --file tests/golden/bfs_test.mtx --undirected --src 0 --validate --steps 3 --warmup 1 > $O/bench_file.log 2>&1
echo "bench file rc=$?"; tail -1 $O/bench_file.log | cut -c1-400
