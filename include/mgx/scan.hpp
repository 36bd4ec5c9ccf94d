// mgx/scan.hpp -- device-wide exclusive transform-scan and stable transform-compact.
//
// These replace the two moderngpu primitives the reference's operators lean on:
//   transform_scan<int>(f, n, out, plus_t<int>(), total, ctx)    advance.hxx:40, neighborhood.hxx:35
//   transform_compact(n, ctx).upsweep(pred) / .downsweep(emit)   filter.hxx:18-29, advance.hxx:93-104
// Contract kept: the compaction predicate is evaluated EXACTLY ONCE per element (functors
// such as sssp/pr cond_filter have side effects) and output order == input order.
//
// Shape: SINGLE PASS (decoupled look-back) with a tile of 2048 items per 256-thread workgroup; the grand total is left
// in scratch and stored into the context's pinned mailbox by the kernel itself (no read-back copy).
// No allocation: partials and predicate bitmasks live in the context's scratch arena, the tiles' status words in the
// context's look-back array.
#pragma once
#include "runtime.hpp"
#include "wave.hpp"

namespace mgx {

constexpr int SCAN_ITEMS = 8;
constexpr int SCAN_TILE = BLOCK * SCAN_ITEMS;   // 2048

inline long long scan_num_tiles(long long n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }
__device__ __forceinline__ long long scan_num_tiles_dev(long long n) { return (n + SCAN_TILE - 1) / SCAN_TILE; }
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

// a count for the host: the value into the pinned mailbox, then -- system scope, release -- the sequence number the host spins on
__device__ __forceinline__ void deliver_count(long long* mailbox, long long value, long long seq) {
  mailbox[0] = value;
  __threadfence_system();
  __hip_atomic_store(mailbox + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- pass 2: one workgroup turns the partials into exclusive prefixes --------------------
// partials[ntiles] receives the grand total.
// total_out: device-visible (the pinned mailbox: the host reads it after the stream, no read-back copy) or NULL.
__global__ __launch_bounds__(BLOCK) void k_scan_partials(long long* __restrict__ partials, long long ntiles,
                                                          long long* __restrict__ total_out, long long seq = 0) {
  __shared__ long long sm[WAVES_PER_BLOCK + 1];
  long long carry = 0;
  for (long long base = 0; base < ntiles; base += BLOCK * SCAN_ITEMS) {     // 8 consecutive partials per thread
    const long long i0 = base + (long long)threadIdx.x * SCAN_ITEMS;
    long long x[SCAN_ITEMS];
    long long mine = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { x[k] = (i0 + k < ntiles) ? partials[i0 + k] : 0; mine += x[k]; }
    long long tot;
    long long run = carry + block_exclusive_sum(mine, sm, &tot);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { if (i0 + k < ntiles) partials[i0 + k] = run; run += x[k]; }
    carry += tot;
  }
  if (threadIdx.x == 0) {
    partials[ntiles] = carry;
    if (total_out) deliver_count(total_out, carry, seq);
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

// ---- single pass: decoupled look-back ------------------------------------------------------------------------------
// One kernel instead of three (tile sums, scan of the partials by ONE workgroup -- 13 us per call, measured --, rescan):
// a tile publishes its aggregate as soon as it has it and then looks back over its predecessors' status words until it
// meets one that already carries an inclusive prefix.  A status word is ONE 64-bit store (agent scope, write-through):
// [epoch : 30][state : 2][value : 32], state 1 = aggregate, 2 = inclusive prefix -- no fence between value and flag
// needed, and because every launch uses a fresh epoch the array is never cleared.  Tiles take their index from a ticket
// counter in launch order, so a tile only ever waits for tiles that started before it.  Values are 32-bit: the
// operators' counts are ints (graph.hxx:19-26).
constexpr unsigned long long LB_AGGREGATE = 1ull, LB_PREFIX = 2ull;
constexpr long long SCAN_LOOKBACK_MAX_TILES = 256;      // above: reduce-then-scan (three launches)
__device__ __forceinline__ unsigned long long lb_word(unsigned epoch, unsigned long long state, unsigned value) {
  return ((unsigned long long)epoch << 34) | (state << 32) | (unsigned long long)value;
}
// exclusive prefix of tile `tile` (> 0): called by the first wave of the workgroup; every lane gets the result
__device__ __forceinline__ unsigned lb_look_back(const unsigned long long* status, long long tile, unsigned epoch) {
  const int lane = lane_id();
  unsigned run = 0;
  long long hi = tile;                      // predecessors [hi - 64, hi) are inspected next, lane 0 the nearest
  for (;;) {
    const long long j = hi - 1 - lane;
    unsigned long long w = 0;
    bool ready = true;                      // lanes before tile 0: a prefix of 0
    unsigned long long state = LB_PREFIX;
    unsigned val = 0;
    if (j >= 0) {
      w = __hip_atomic_load(status + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      state = (w >> 32) & 3ull;
      ready = (unsigned)(w >> 34) == epoch && state != 0ull;
      val = (unsigned)w;
    }
    // the window is usable up to (and including) the nearest PREFIX once every nearer word is ready
    const u64 not_ready = __ballot(!ready);
    const u64 is_prefix = __ballot(ready && state == LB_PREFIX);
    const int first_prefix = is_prefix ? __ffsll((long long)is_prefix) - 1 : WAVE;
    const int first_wait = not_ready ? __ffsll((long long)not_ready) - 1 : WAVE;
    if (first_wait < first_prefix && first_wait < WAVE) {     // a nearer tile has not published yet: poll again
      __builtin_amdgcn_s_sleep(1);
      continue;
    }
    const int take = first_prefix < WAVE ? first_prefix + 1 : WAVE;     // lanes [0, take) contribute
    unsigned mine = lane < take ? val : 0u;
    run += wave_sum(mine);
    if (first_prefix < WAVE) return run;
    hi -= WAVE;
  }
}

// out[i] = exclusive sum of f over [0, i); *total_dev and *total_host (pinned, device-visible; may be NULL) = the sum
template <typename F>
__global__ __launch_bounds__(BLOCK) void k_scan_lookback(F f, long long n, int* __restrict__ out, unsigned long long* status,
                                                         unsigned* ticket, unsigned ticket_base, unsigned epoch,
                                                         long long* total_dev, long long* total_host, long long seq) {
  __shared__ unsigned sm[WAVES_PER_BLOCK + 1];
  __shared__ unsigned s_tile, s_prefix;
  if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u) - ticket_base;
  __syncthreads();
  const long long tile = (long long)s_tile;
  const long long base = tile * SCAN_TILE + (long long)threadIdx.x * SCAN_ITEMS;     // blocked: 8 consecutive items per thread
  unsigned x[SCAN_ITEMS];
  unsigned mine = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    x[k] = (base + k < n) ? (unsigned)f(base + k) : 0u;
    mine += x[k];
  }
  unsigned tot;
  const unsigned ex = block_exclusive_sum(mine, sm, &tot);
  if (threadIdx.x < WAVE) {
    unsigned prefix = 0;
    if (tile > 0) {
      if (threadIdx.x == 0) __hip_atomic_store(status + tile, lb_word(epoch, LB_AGGREGATE, tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      prefix = lb_look_back(status, tile, epoch);
    }
    if (threadIdx.x == 0) {
      __hip_atomic_store(status + tile, lb_word(epoch, LB_PREFIX, prefix + tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_prefix = prefix;
      if (tile == scan_num_tiles_dev(n) - 1) {
        if (total_dev) *total_dev = (long long)(prefix + tot);
        if (total_host) deliver_count(total_host, (long long)(prefix + tot), seq);
      }
    }
  }
  __syncthreads();
  unsigned run = s_prefix + ex;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    if (base + k < n) out[base + k] = (int)run;
    run += x[k];
  }
}

// Exclusive plus-scan of f(0..n-1) into out[]; the total lands in scratch and, if `host_total` is set, in the pinned
// mailbox: the host waits for the stream and reads it (this is the blocking 8-byte copy of advance.hxx:43, without the
// copy).  Returns the device pointer of the 64-bit total.
template <typename F>
inline long long* transform_scan(F f, long long n, int* out, standard_context_t& ctx, long long* host_total) {
  const long long ntiles = scan_num_tiles(n);
  ++ctx.scratch_epoch;
  long long* const partials = (long long*)ctx.scratch;
  long long* const d_total = partials + ntiles;
  if ((size_t)ntiles > ctx.lookback_tiles || (size_t)(ntiles + 2) * sizeof(long long) > ctx.scratch_bytes)
    throw mgx_error(MGX_E_INVALID, "scan: scratch arena too small (reserve_scratch was not called for this size)");
  hipStream_t st = ctx.stream();
  long long* const host_slot = host_total ? ctx.mailbox : (long long*)nullptr;
  const long long seq = host_total ? ++ctx.mailbox_seq : 0;
  if (n > 0 && ntiles <= SCAN_LOOKBACK_MAX_TILES) {
    // every tile resident at once: a tile looks back over aggregates that are published as soon as their tiles have
    // summed up -- a handful of polls.  (With tens of thousands of tiles the look-back of each one crosses the ~2000
    // tiles in flight, window by window at ~1 us per dependent cross-XCD poll: the RMAT-22 operator-path traversal
    // went from 2.5 to 2.9 ms with the single pass everywhere.)
    const unsigned epoch = ctx.next_lookback_epoch();
    hipLaunchKernelGGL(k_scan_lookback<F>, dim3((unsigned)ntiles), dim3(BLOCK), 0, st, f, n, out, ctx.lookback_status,
                       ctx.lookback_ticket, ctx.lookback_ticket_base, epoch, d_total, host_slot, seq);
    ctx.lookback_ticket_base += (unsigned)ntiles;
  } else if (n > 0) {
    hipLaunchKernelGGL(k_scan_tile_sums<F>, dim3((unsigned)ntiles), dim3(BLOCK), 0, st, f, n, partials);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(BLOCK), 0, st, partials, ntiles, host_slot, seq);
    hipLaunchKernelGGL(k_scan_downsweep<F>, dim3((unsigned)ntiles), dim3(BLOCK), 0, st, f, n, partials, out);
  } else {
    MGX_HIP(hipMemsetAsync(d_total, 0, sizeof(long long), st));
    ctx.mailbox[0] = 0;
  }
  if (host_total) {
    if (n > 0) {
      MGX_CHECK_LAUNCH("scan: kernel launch");
      ctx.mailbox_wait(seq);        // (the rescan of the three-launch shape may still be running: the total does not wait for it)
    }
    *host_total = n > 0 ? ctx.mailbox[0] : 0;
  }
  return d_total;
}

// ---- stable compaction --------------------------------------------------------------------
// upsweep: evaluate pred(i) once, remember the answers as one 64-bit ballot per wave-row
// (element i <-> bit i%64 of word i/64), count per tile -- and, in the same launch, turn the tile counts into
// exclusive prefixes by decoupled look-back (as k_scan_lookback: no one-workgroup pass over the partials, no
// read-back copy: the last tile stores the kept count in the pinned mailbox).
// FROM_BITS: the ballot words are there already (an advance left them: lbs.hpp, k_transform_lbs_keep) -- count them, pred is not called.
template <typename P, bool FROM_BITS = false>
__global__ __launch_bounds__(BLOCK) void k_compact_upsweep(P pred, long long n, u64* __restrict__ bits,
                                                            long long* __restrict__ partials, unsigned long long* status,
                                                            unsigned* ticket, unsigned ticket_base, unsigned epoch,
                                                            long long* total_host, long long seq) {
  __shared__ int sm[WAVES_PER_BLOCK + 1];
  __shared__ unsigned s_tile;
  // (tickets only where tiles wait for each other: ONE hot counter serves ~83 M returning adds a second, 57 000 tiles
  //  of a big level's compaction spent 0.6 ms queueing for theirs)
  if (status) {
    if (threadIdx.x == 0) s_tile = atomicAdd(ticket, 1u) - ticket_base;
    __syncthreads();
  }
  const long long tile = status ? (long long)s_tile : (long long)blockIdx.x;
  const long long base = tile * SCAN_TILE;
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < SCAN_ITEMS; ++k) {
    const long long i = base + k * BLOCK + threadIdx.x;
    if constexpr (FROM_BITS) {
      if (lane_id() == 0 && i < n) cnt += __popcll(bits[i / 64]);      // (i: this wave-row's first element, a multiple of 64)
    } else {
      bool keep = false;
      if (i < n) keep = pred(i);
      const u64 m = __ballot(keep);
      if (lane_id() == 0 && (base + k * BLOCK + (threadIdx.x / WAVE) * WAVE) < n) {
        bits[i / 64] = m;   // i is this wave-row's first element, a multiple of 64
        cnt += __popcll(m);
      }
    }
  }
  if (lane_id() == 0) sm[threadIdx.x / WAVE] = cnt;
  __syncthreads();
  if (threadIdx.x < WAVE) {
    unsigned t = 0;
    for (int w = 0; w < WAVES_PER_BLOCK; ++w) t += (unsigned)sm[w];
    if (!status) {                       // many tiles: k_scan_partials turns the counts into prefixes
      if (threadIdx.x == 0) partials[tile] = (long long)t;
      return;
    }
    unsigned prefix = 0;
    if (tile > 0) {
      if (threadIdx.x == 0) __hip_atomic_store(status + tile, lb_word(epoch, LB_AGGREGATE, t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      prefix = lb_look_back(status, tile, epoch);
    }
    if (threadIdx.x == 0) {
      __hip_atomic_store(status + tile, lb_word(epoch, LB_PREFIX, prefix + t), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      partials[tile] = (long long)prefix;
      if (tile == scan_num_tiles_dev(n) - 1) {
        partials[tile + 1] = (long long)(prefix + t);
        if (total_host) deliver_count(total_host, (long long)(prefix + t), seq);
      }
    }
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
  struct no_pred_t { __device__ bool operator()(long long) const { return false; } };
  // the ballots are in `bits` already (transform_lbs_keep): counts, prefixes and the kept total only
  long long upsweep_from_bits() { return upsweep_impl<no_pred_t, true>(no_pred_t()); }
  // returns the kept count (blocking 8-byte read-back, as mgpu's upsweep does)
  template <typename P>
  long long upsweep(P pred) { return upsweep_impl<P, false>(pred); }
  template <typename P, bool FROM_BITS>
  long long upsweep_impl(P pred) {
    hipStream_t st = ctx.stream();
    ++ctx.scratch_epoch;
    if (n <= 0) return 0;
    if ((size_t)ntiles > ctx.lookback_tiles) throw mgx_error(MGX_E_INVALID, "compact: scratch arena too small");
    const bool single = ntiles <= SCAN_LOOKBACK_MAX_TILES;       // (see transform_scan)
    const unsigned epoch = single ? ctx.next_lookback_epoch() : 0u;
    const long long seq = ++ctx.mailbox_seq;
    hipLaunchKernelGGL((k_compact_upsweep<P, FROM_BITS>), dim3((unsigned)ntiles), dim3(BLOCK), 0, st, pred, n, bits, partials,
                       single ? ctx.lookback_status : (unsigned long long*)nullptr, ctx.lookback_ticket, ctx.lookback_ticket_base,
                       epoch, ctx.mailbox, seq);
    if (single) ctx.lookback_ticket_base += (unsigned)ntiles;
    if (!single) hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(BLOCK), 0, st, partials, ntiles, ctx.mailbox, seq);
    MGX_CHECK_LAUNCH("compact: kernel launch");
    ctx.mailbox_wait(seq);
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
