#!/bin/bash
# after the lab clean-up: the -m gpu suite on the product and on the lab library, smoke, the three bench lines
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_suite2; rm -rf $O; mkdir -p $O; cd $R
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
MGX_LIB=mini_amd/libmgx_lab.so timeout 2000 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu > $O/pytest_lab.txt 2>&1; tail -3 $O/pytest_lab.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for a in push sssp pr; do timeout 600 python bench.py --mode $a --no-cpu-baseline > $O/bench_$a.json 2> $O/bench_$a.err; cut -c1-300 $O/bench_$a.json; done
