#!/bin/bash
# ASan + UBSan run of the product's host-only entry points (MatrixMarket loader, binary CSR cache: what mgx_load_mtx,
# mgx_load_mtx_csc, mgx_graph_save_csr and mgx_graph_load_csr wrap): the sanitizers instrument the HOST side only
# (-Xarch_host; the GPU pool has no device ASan), no HIP call is made, no GPU is needed.
# usage: bash tools/host_asan/run.sh        (exits non-zero on a sanitizer finding or a failed expectation)
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"
WORK=$(mktemp -d /tmp/mgx_host_asan.XXXXXX)
trap 'rm -rf "$WORK"' EXIT
timeout 600 ${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 --cuda-host-only -O1 -g -std=c++17 -Wno-unused-value -I"$ROOT/include" \
    -Xarch_host -fsanitize=address,undefined -Xarch_host -fno-sanitize-recover=undefined -Xarch_host -fno-omit-frame-pointer \
    -fsanitize=address,undefined "$ROOT/tools/host_asan/host_io_asan.hip" -o "$WORK/host_io_asan"
mkdir -p "$WORK/scratch"
ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1 timeout 600 "$WORK/host_io_asan" "$ROOT/tests/golden" "$WORK/scratch" | grep -v "^Error reading"
