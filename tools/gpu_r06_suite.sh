#!/bin/bash
# the whole -m gpu suite + smoke + the default bench line
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06_suite; rm -rf $O; mkdir -p $O; cd $R
timeout 3000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
