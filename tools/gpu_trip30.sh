#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/gpu_trace.sh > /dev/null 2>&1
python3 tools/trace_one.py gpurun_out/trace_quick 10
