// mgx/scan.hpp -- device-wide exclusive transform-scan and stable transform-compact.
//
// These replace the two moderngpu primitives the reference's operators lean on:
//   transform_scan<int>(f, n, out, plus_t<int>(), total, ctx)    advance.hxx:40, neighborhood.hxx:35
//   transform_compact(n, ctx).upsweep(pred) / .downsweep(emit)   filter.hxx:18-29, advance.hxx:93-104
// Contract kept: the compaction predicate is evaluated EXACTLY ONCE per element (functors
// such as sssp/pr cond_filter have side effects) and output order == input order.
//
// Shape: reduce-then-scan with a tile of 2048 items per 256-thread workgroup, 8 items per
// lane strided by the block so every global access is a full 256 B wave transaction.
// Tile partials are 64-bit; the one-workgroup middle pass scans them and leaves the grand
// total in scratch (and in the context's pinned mailbox when the caller asks for it).
// No allocation: partials and predicate bitmasks live in the context's scratch arena.
#pragma once
#include "runtime.hpp"
#include "wave.hpp"

namespace mgx {

constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = BLOCK * SCAN_ITEMS;   // 2048

inline long long scan_num_tiles(long long n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }
// scratch bytes needed by a scan / compaction over n items
inline size_t scan_scratch_bytes(long long n) {
  return (size_t)(scan_num_tiles(n) + 2) * sizeof(long long)      // partials + total
         + (size_t)((n + 63) / 64 + 4) * sizeof(u64) + 256;       // predicate bits
}

// ---- pass 1: per-tile sums -----------------------------------------------------------
template <typename F>
__global__ __launch_bounds__(BLOCK) void k_scan_tile_sums(F f, long long n, long long* __restrict__ partials) {
  __shared__ long long sm[WAVES_PER_BLOCK + 1];
  const long long base = (long long)blockIdx.x * SCAN_TILE;
  long long s = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    const long long i = base + k * BLOCK + threadIdx.x;
    if (i < n) s += (long long)f(i);
  }
  s = wave_sum(s);
  if (lane_id() == 0) sm[threadIdx.x / WAVE] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    long long t = 0;
    for (int w = 0; w < WAVES_PER_BLOCK; ++w) t += sm[w];
    partials[blockIdx.x] = t;
  }
}

// ---- pass 2: one workgroup turns the partials into exclusive prefixes --------------------
// partials[ntiles] receives the grand total.
__global__ __launch_bounds__(BLOCK) void k_scan_partials(long long* __restrict__ partials, long long ntiles,
                                                          long long* __restrict__ total_out) {
  __shared__ long long sm[WAVES_PER_BLOCK + 1];
  long long carry = 0;
  for (long long base = 0; base < ntiles; base += BLOCK) {
    const long long i = base + threadIdx.x;
    long long x = (i < ntiles) ? partials[i] : 0;
    long long tot;
    long long ex = block_exclusive_sum(x, sm, &tot);
    if (i < ntiles) partials[i] = carry + ex;
    carry += tot;
  }
  if (threadIdx.x == 0) {
    partials[ntiles] = carry;
    if (total_out) *total_out = carry;
  }
}

// ---- pass 3: rescan each tile with its base --------------------------------------------
template <typename F>
__global__ __launch_bounds__(BLOCK) void k_scan_downsweep(F f, long long n, const long long* __restrict__ partials,
                                                           int* __restrict__ out) {
  __shared__ int sm[WAVES_PER_BLOCK + 1];
  const long long base = (long long)blockIdx.x * SCAN_TILE;
  int run = (int)partials[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    const long long i = base + k * BLOCK + threadIdx.x;
    int x = (i < n) ? (int)f(i) : 0;
    int tot;
    int ex = block_exclusive_sum(x, sm, &tot);
    if (i < n) out[i] = run + ex;
    run += tot;
  }
}

// Exclusive plus-scan of f(0..n-1) into out[]; the total lands in scratch and, if
// `host_total` is set, is read back (this is the blocking 8-byte copy of advance.hxx:43).
// Returns the device pointer of the 64-bit total.
template <typename F>
inline long long* transform_scan(F f, long long n, int* out, standard_context_t& ctx, long long* host_total) {
  long long* partials = (long long*)ctx.scratch;
  const long long ntiles = scan_num_tiles(n);
  if ((size_t)(ntiles + 2) * sizeof(long long) > ctx.scratch_bytes)
    throw mgx_error(MGX_E_INVALID, "scan: scratch arena too small (reserve_scratch was not called for this size)");
  hipStream_t st = ctx.stream();
  if (n > 0) {
    hipLaunchKernelGGL(k_scan_tile_sums<F>, dim3((unsigned)ntiles), dim3(BLOCK), 0, st, f, n, partials);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(BLOCK), 0, st, partials, ntiles, (long long*)nullptr);
    hipLaunchKernelGGL(k_scan_downsweep<F>, dim3((unsigned)ntiles), dim3(BLOCK), 0, st, f, n, partials, out);
  } else {
    MGX_HIP(hipMemsetAsync(partials, 0, sizeof(long long), st));
  }
  long long* d_total = partials + ntiles;
  if (host_total) {
    MGX_HIP(hipMemcpyAsync(ctx.mailbox, d_total, sizeof(long long), hipMemcpyDeviceToHost, st));
    MGX_HIP(hipStreamSynchronize(st));
    *host_total = ctx.mailbox[0];
  }
  return d_total;
}

// ---- stable compaction --------------------------------------------------------------------
// upsweep: evaluate pred(i) once, remember the answers as one 64-bit ballot per wave-row
// (element i <-> bit i%64 of word i/64), count per tile.
template <typename P>
__global__ __launch_bounds__(BLOCK) void k_compact_upsweep(P pred, long long n, u64* __restrict__ bits,
                                                            long long* __restrict__ partials) {
  __shared__ int sm[WAVES_PER_BLOCK + 1];
  const long long base = (long long)blockIdx.x * SCAN_TILE;
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    const long long i = base + k * BLOCK + threadIdx.x;
    bool keep = false;
    if (i < n) keep = pred(i);
    const u64 m = __ballot(keep);
    if (lane_id() == 0 && (base + k * BLOCK + (threadIdx.x / WAVE) * WAVE) < n) {
      bits[i / 64] = m;   // i is this wave-row's first element, a multiple of 64
      cnt += __popcll(m);
    }
  }
  if (lane_id() == 0) sm[threadIdx.x / WAVE] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    int t = 0;
    for (int w = 0; w < WAVES_PER_BLOCK; ++w) t += sm[w];
    partials[blockIdx.x] = t;
  }
}

// downsweep: emit(dest, source) for every kept element, stable.
template <typename E>
__global__ __launch_bounds__(BLOCK) void k_compact_downsweep(E emit, long long n, const u64* __restrict__ bits,
                                                              const long long* __restrict__ partials) {
  __shared__ int wordpre[SCAN_TILE / 64 + 1];
  const long long base = (long long)blockIdx.x * SCAN_TILE;
  const long long word0 = base / 64;
  const long long nwords = (n + 63) / 64;
  // 32 words per tile: wave 0 computes their exclusive popcount prefix
  if (threadIdx.x < WAVE) {
    const int w = threadIdx.x;
    int c = 0;
    if (w < SCAN_TILE / 64 && word0 + w < nwords) c = __popcll(bits[word0 + w]);
    int inc = wave_inclusive_sum(c);
    if (w < SCAN_TILE / 64) wordpre[w] = inc - c;
  }
  __syncthreads();
  const long long tile_base = partials[blockIdx.x];
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    const long long i = base + k * BLOCK + threadIdx.x;
    if (i < n) {
      const int w = (int)(i / 64 - word0);
      const u64 m = bits[word0 + w];
      const int lane = (int)(i & 63);
      if ((m >> lane) & 1ull) {
        const int r = __popcll(m & ((1ull << lane) - 1ull));
        emit(tile_base + wordpre[w] + r, i);
      }
    }
  }
}

struct compact_t {
  standard_context_t& ctx;
  long long n;
  long long ntiles;
  long long* partials;
  u64* bits;
  compact_t(long long count, standard_context_t& c) : ctx(c), n(count), ntiles(scan_num_tiles(count)) {
    if (scan_scratch_bytes(n) > ctx.scratch_bytes)
      throw mgx_error(MGX_E_INVALID, "compact: scratch arena too small (reserve_scratch was not called for this size)");
    partials = (long long*)ctx.scratch;
    bits = (u64*)(((uintptr_t)(partials + ntiles + 2) + 255) & ~(uintptr_t)255);
  }
  // returns the kept count (blocking 8-byte read-back, as mgpu's upsweep does)
  template <typename P>
  long long upsweep(P pred) {
    hipStream_t st = ctx.stream();
    if (n <= 0) return 0;
    hipLaunchKernelGGL(k_compact_upsweep<P>, dim3((unsigned)ntiles), dim3(BLOCK), 0, st, pred, n, bits, partials);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(BLOCK), 0, st, partials, ntiles, (long long*)nullptr);
    MGX_HIP(hipMemcpyAsync(ctx.mailbox, partials + ntiles, sizeof(long long), hipMemcpyDeviceToHost, st));
    MGX_HIP(hipStreamSynchronize(st));
    return ctx.mailbox[0];
  }
  template <typename E>
  void downsweep(E emit) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_compact_downsweep<E>, dim3((unsigned)ntiles), dim3(BLOCK), 0, ctx.stream(), emit, n, bits,
                       partials);
  }
};
inline compact_t transform_compact(long long count, standard_context_t& ctx) { return compact_t(count, ctx); }

}  // namespace mgx
