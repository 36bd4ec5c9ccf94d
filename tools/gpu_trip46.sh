#!/bin/bash
cd $GRAFT_REPO_ROOT
ulimit -c 0
timeout 600 python -u tools/pr_bench.py 2>&1 | grep -v amdgpu.ids | tail -12
