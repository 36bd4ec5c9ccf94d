#!/bin/bash
# Decomposition of the stream kernel at one level with the instrumented build (MGX_BFS_FLAGS = level<<8 | bits).
ulimit -c 0
mkdir -p gpurun_out
L=${1:-3}
for bits in ${BITS:-0 16 2 1 3}; do
  F=$(( (L<<8) | bits ))
  echo "== flags level $L bits $bits" >> gpurun_out/diag.log
  MGX_BFS_FLAGS=$F timeout 300 python3 tools/bfs_levels.py --runs 1 2>&1 | grep -E "level  $L|slots|stream|diag" >> gpurun_out/diag.log
done
cat gpurun_out/diag.log
