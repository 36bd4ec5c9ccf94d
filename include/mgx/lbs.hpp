// mgx/lbs.hpp -- load-balanced search over a scanned segment list, and its segmented reduce.
//
// Replaces moderngpu's transform_lbs / lbs_segreduce (reference call sites advance.hxx:62,157;
// neighborhood.hxx:58).  Work item idx in [0,count) belongs to the segment seg with
// segments[seg] <= idx < segments[seg+1] (segments[] is the exclusive degree scan kept in
// graph_device_t::d_scanned_row_offsets); rank = idx - segments[seg].  Empty segments own
// nothing and may appear in any number.
//
// Shape (gfx950): one 256-thread workgroup per tile of LBS_TILE consecutive work items, so
// the dependent col_indices loads of one CSR row are consecutive lanes of a wave.  Each
// tile finds its first/last segment with a wave-cooperative 64-ary search (4 probes for
// 2^24 segments instead of 24 dependent loads), stages that slice of the scan in LDS and
// every lane resolves its own item with a binary search in LDS.  Slices longer than the LDS
// window (only possible with long runs of empty segments) fall back to searching global memory.
#pragma once
#include "runtime.hpp"
#include "wave.hpp"

namespace mgx {

constexpr int LBS_ITEMS = 4;
constexpr int LBS_TILE = BLOCK * LBS_ITEMS;     // 1024 work items per workgroup
constexpr int LBS_WINDOW = LBS_TILE + 64;       // segment offsets staged in LDS

// number of j in [0,n) with a[j] <= key (a sorted ascending), a in LDS or global
template <typename A>
__device__ __forceinline__ int upper_bound_small(const A& a, int n, int key) {
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] <= key) lo = mid + 1; else hi = mid;
  }
  return lo;
}

struct lbs_tile_t {
  long long seg_lo;     // first segment overlapping the tile
  int nseg;             // segments staged (0 => global fallback)
  long long seg_hi;     // last segment overlapping the tile
};

// Locate the tile's segment slice and stage it.  Must be called by all BLOCK threads.
// s_off must hold LBS_WINDOW ints.
__device__ __forceinline__ lbs_tile_t lbs_stage_tile(const int* __restrict__ segments, long long num_segments,
                                                     long long first, long long last /*inclusive*/,
                                                     int* s_off, long long* s_bounds) {
  const int wave = threadIdx.x / WAVE;
  if (wave == 0) {
    long long ub = wave_upper_bound(segments, num_segments, (int)first);
    if (lane_id() == 0) s_bounds[0] = ub - 1;
  } else if (wave == 1) {
    long long ub = wave_upper_bound(segments, num_segments, (int)last);
    if (lane_id() == 0) s_bounds[1] = ub - 1;
  }
  __syncthreads();
  lbs_tile_t t;
  t.seg_lo = s_bounds[0];
  t.seg_hi = s_bounds[1];
  const long long span = t.seg_hi - t.seg_lo + 1;
  t.nseg = (span <= LBS_WINDOW) ? (int)span : 0;
  if (t.nseg) {
    for (int j = threadIdx.x; j < t.nseg; j += BLOCK) s_off[j] = segments[t.seg_lo + j];
  }
  __syncthreads();
  return t;
}

// f(idx, seg, rank) for every work item
template <typename F>
__global__ __launch_bounds__(BLOCK) void k_transform_lbs(F f, long long count, const int* __restrict__ segments,
                                                          long long num_segments) {
  __shared__ int s_off[LBS_WINDOW];
  __shared__ long long s_bounds[2];
  for (long long tile = blockIdx.x; tile * LBS_TILE < count; tile += gridDim.x) {
    const long long first = tile * LBS_TILE;
    const long long last = (first + LBS_TILE < count ? first + LBS_TILE : count) - 1;
    lbs_tile_t t = lbs_stage_tile(segments, num_segments, first, last, s_off, s_bounds);
#pragma unroll
    for (int k = 0; k < LBS_ITEMS; ++k) {
      const long long idx = first + k * BLOCK + threadIdx.x;
      if (idx <= last) {
        long long seg;
        int start;
        if (t.nseg) {
          const int j = upper_bound_small(s_off, t.nseg, (int)idx) - 1;
          seg = t.seg_lo + j;
          start = s_off[j];
        } else {
          const int* a = segments + t.seg_lo;
          const int j = upper_bound_small(a, (int)(t.seg_hi - t.seg_lo + 1), (int)idx) - 1;
          seg = t.seg_lo + j;
          start = a[j];
        }
        f((int)idx, (int)seg, (int)idx - start);
      }
    }
    __syncthreads();   // s_off / s_bounds are reused by the next tile
  }
}

// The same enumeration with f returning a bool per work item: the answers of 64 consecutive items -- one wave instruction: item
// idx sits in lane idx % 64 of the wave that handles items [idx & ~63, + 64) -- go to bits[idx / 64] as one ballot word, the
// layout the stable compaction reads (scan.hpp: compact_t).  An advance whose functor says that its filter test depends on the
// slot's value alone evaluates that test here, while the value is in a register: the filter behind it never reads the raw
// frontier to decide, only to copy the survivors (gunrock/advance.hxx, filter.hxx).
template <typename F>
__global__ __launch_bounds__(BLOCK) void k_transform_lbs_keep(F f, long long count, const int* __restrict__ segments,
                                                               long long num_segments, u64* __restrict__ bits) {
  __shared__ int s_off[LBS_WINDOW];
  __shared__ long long s_bounds[2];
  for (long long tile = blockIdx.x; tile * LBS_TILE < count; tile += gridDim.x) {
    const long long first = tile * LBS_TILE;
    const long long last = (first + LBS_TILE < count ? first + LBS_TILE : count) - 1;
    lbs_tile_t t = lbs_stage_tile(segments, num_segments, first, last, s_off, s_bounds);
#pragma unroll
    for (int k = 0; k < LBS_ITEMS; ++k) {
      const long long idx = first + k * BLOCK + threadIdx.x;
      bool keep = false;
      if (idx <= last) {
        long long seg;
        int start;
        if (t.nseg) {
          const int j = upper_bound_small(s_off, t.nseg, (int)idx) - 1;
          seg = t.seg_lo + j;
          start = s_off[j];
        } else {
          const int* a = segments + t.seg_lo;
          const int j = upper_bound_small(a, (int)(t.seg_hi - t.seg_lo + 1), (int)idx) - 1;
          seg = t.seg_lo + j;
          start = a[j];
        }
        keep = f((int)idx, (int)seg, (int)idx - start);
      }
      const u64 m = __ballot(keep);
      const long long row_first = idx - lane_id();             // a multiple of 64 (tiles are 1024 items, rows 64)
      if (lane_id() == 0 && row_first <= last) bits[row_first / 64] = m;
    }
    __syncthreads();   // s_off / s_bounds are reused by the next tile
  }
}

template <typename F>
inline void transform_lbs_keep(F f, long long count, const int* segments, long long num_segments, u64* bits,
                               standard_context_t& ctx) {
  if (count <= 0) return;
  long long tiles = (count + LBS_TILE - 1) / LBS_TILE;
  const long long cap = (long long)ctx.num_cus * 32;
  hipLaunchKernelGGL(k_transform_lbs_keep<F>, dim3((unsigned)(tiles < cap ? tiles : cap)), dim3(BLOCK), 0, ctx.stream(), f,
                     count, segments, num_segments, bits);
}

template <typename F>
inline void transform_lbs(F f, long long count, const int* segments, long long num_segments,
                          standard_context_t& ctx) {
  if (count <= 0) return;
  long long tiles = (count + LBS_TILE - 1) / LBS_TILE;
  const long long cap = (long long)ctx.num_cus * 32;
  hipLaunchKernelGGL(k_transform_lbs<F>, dim3((unsigned)(tiles < cap ? tiles : cap)), dim3(BLOCK), 0, ctx.stream(), f,
                     count, segments, num_segments);
}

// ---- segmented reduce over the same enumeration ----------------------------------------------
// reduced[seg] = op-fold of f(idx, seg, rank) over the segment's items, identity for empty segments.  Deterministic (fixed
// association order, no atomics on values); a segment that spans tiles leaves one partial per tile (carry_val / carry_seg), and the
// fix-up kernel gives every such segment to the thread of its opening tile, which walks the following tiles' continuation slots
// in order.

// ---- the same reduce, second generation: wave-level segmented scans instead of LDS-staged serial folds ------------
// In the first generation (k_lbs_segreduce: in this file's history until round 6) one lane folds a run of up to 32 items while its neighbours idle (RMAT rows average 32 items),
// and every value makes a round trip through LDS.  Here a wave holds 64 CONSECUTIVE items of the tile in registers
// (item = k * BLOCK + thread, so wave w of slab k holds items [64 (4k + w), +64)) and folds them with a segmented
// inclusive scan over the lanes (6 shuffle steps, earlier items on the left: deterministic).  A run that lies
// strictly inside a slab is complete and written at once; the first and the last run of a slab may continue in
// the neighbouring slabs: their partials go to LDS (16 slabs x 2) and one thread stitches them in slab order,
// handing complete segments to reduced[] and tile-spanning ones to the same carry slots k_segreduce_fixup reads.
template <typename T, typename F, typename Op>
__global__ __launch_bounds__(BLOCK) void k_lbs_segreduce2(F f, long long count, const int* __restrict__ segments,
                                                           long long num_segments, T* __restrict__ reduced, Op op,
                                                           T identity, T* __restrict__ carry_val,
                                                           long long* __restrict__ carry_seg) {
  constexpr int NSLAB = LBS_TILE / WAVE;          // 16
  __shared__ int s_off[LBS_WINDOW];
  __shared__ long long s_bounds[2];
  __shared__ T s_hval[NSLAB], s_tval[NSLAB];
  __shared__ long long s_hseg[NSLAB], s_tseg[NSLAB];
  const long long tile = blockIdx.x;
  const long long first = tile * LBS_TILE;
  const long long last = (first + LBS_TILE < count ? first + LBS_TILE : count) - 1;
  lbs_tile_t t = lbs_stage_tile(segments, num_segments, first, last, s_off, s_bounds);
  const int lane = lane_id();
  const int wave = threadIdx.x / WAVE;
  if (threadIdx.x < NSLAB) { s_hseg[threadIdx.x] = -1; s_tseg[threadIdx.x] = -1; }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < LBS_ITEMS; ++k) {
    const int slab = k * WAVES_PER_BLOCK + wave;
    const long long idx = first + k * BLOCK + threadIdx.x;
    const bool valid = idx <= last;
    long long seg = -1;
    T x = identity;
    if (valid) {
      int start;
      if (t.nseg) {
        const int j = upper_bound_small(s_off, t.nseg, (int)idx) - 1;
        seg = t.seg_lo + j;
        start = s_off[j];
      } else {
        const int* a = segments + t.seg_lo;
        const int j = upper_bound_small(a, (int)(t.seg_hi - t.seg_lo + 1), (int)idx) - 1;
        seg = t.seg_lo + j;
        start = a[j];
      }
      x = f((int)idx, (int)seg, (int)idx - start);
    }
    const int sg = (int)(seg - t.seg_lo);                    // -1 - seg_lo for invalid lanes: never equals a valid one
    // segmented inclusive scan: lane i ends up with the fold of its run's items up to itself
#pragma unroll
    for (int d = 1; d < WAVE; d <<= 1) {
      const T y = __shfl_up(x, d, WAVE);
      const int sy = __shfl_up(sg, d, WAVE);
      if (lane >= d && sy == sg && valid) x = op(y, x);
    }
    const int sg_next = __shfl_down(sg, 1, WAVE);
    const bool valid_next = __shfl_down((int)valid, 1, WAVE) != 0;
    const bool is_tail = valid && (lane == WAVE - 1 || !valid_next || sg_next != sg);
    const int sg_first = __shfl(sg, 0, WAVE);                // lane 0 of a slab that has items is valid
    const u64 vmask = __ballot(valid);
    const int last_lane = vmask ? 63 - __builtin_clzll(vmask) : -1;
    const int sg_last = __shfl(sg, last_lane < 0 ? 0 : last_lane, WAVE);
    if (is_tail) {
      const bool in_first = sg == sg_first, in_last = sg == sg_last;
      if (!in_first && !in_last) reduced[seg] = x;           // strictly inside the slab: complete
      if (in_first) { s_hseg[slab] = seg; s_hval[slab] = x; }
      if (in_last) { s_tseg[slab] = seg; s_tval[slab] = x; }
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // where a stitched run goes: straight to reduced[] if the segment lies inside this tile, else a carry
    auto flush = [&](long long seg, T acc) {
      if (seg < 0) return;
      const long long seg_begin = segments[seg];
      const long long seg_end = (seg + 1 < num_segments) ? (long long)segments[seg + 1] : count;
      const bool opens_here = seg_begin >= first;
      const bool closes_here = seg_end - 1 <= last;
      if (opens_here && closes_here) {
        reduced[seg] = acc;
      } else {
        const int slot = opens_here ? 1 : 0;     // 0: continues a segment begun earlier, 1: left open
        carry_val[tile * 2 + slot] = acc;
        carry_seg[tile * 2 + slot] = seg;
      }
    };
    long long acc_seg = -1;
    T acc = identity;
    for (int sl = 0; sl < NSLAB; ++sl) {
      const long long hs = s_hseg[sl];
      if (hs < 0) continue;                      // slab without items (end of the data)
      if (hs == acc_seg) acc = op(acc, s_hval[sl]);
      else { flush(acc_seg, acc); acc_seg = hs; acc = s_hval[sl]; }
      const long long ts = s_tseg[sl];
      if (ts != hs) { flush(acc_seg, acc); acc_seg = ts; acc = s_tval[sl]; }
    }
    flush(acc_seg, acc);
  }
}

// empty segments get the identity (they own no item so no tile ever writes them)
template <typename T>
__global__ __launch_bounds__(BLOCK) void k_segreduce_fill_empty(const int* __restrict__ segments, long long num_segments,
                                                                long long count, T* __restrict__ reduced, T identity) {
  long long s = (long long)blockIdx.x * BLOCK + threadIdx.x;
  const long long stride = (long long)gridDim.x * BLOCK;
  for (; s < num_segments; s += stride) {
    const long long b = segments[s];
    const long long e = (s + 1 < num_segments) ? (long long)segments[s + 1] : count;
    if (e == b) reduced[s] = identity;
  }
}

// fold the per-tile carries: the thread of the tile that OPENED a segment (slot 1) walks the
// continuation slots (slot 0) of the following tiles, in tile order.
template <typename T, typename Op>
__global__ __launch_bounds__(BLOCK) void k_segreduce_fixup(long long ntiles, const T* __restrict__ carry_val,
                                                            const long long* __restrict__ carry_seg,
                                                            T* __restrict__ reduced, Op op) {
  long long tile = (long long)blockIdx.x * BLOCK + threadIdx.x;
  const long long stride = (long long)gridDim.x * BLOCK;
  for (; tile < ntiles; tile += stride) {
    const long long sg = carry_seg[tile * 2 + 1];
    if (sg < 0) continue;
    T acc = carry_val[tile * 2 + 1];
    for (long long u = tile + 1; u < ntiles && carry_seg[u * 2] == sg; ++u) acc = op(acc, carry_val[u * 2]);
    reduced[sg] = acc;
  }
}

inline size_t segreduce_scratch_bytes(long long count, size_t value_size) {
  const long long tiles = (count + LBS_TILE - 1) / LBS_TILE + 1;
  return (size_t)tiles * 2 * (value_size + sizeof(long long)) + 512;
}

template <typename T, typename F, typename Op>
inline void lbs_segreduce(F f, long long count, const int* segments, long long num_segments, T* reduced, Op op,
                          T identity, standard_context_t& ctx) {
  hipStream_t st = ctx.stream();
  if (num_segments > 0)
    hipLaunchKernelGGL(k_segreduce_fill_empty<T>, dim3(grid_for(num_segments)), dim3(BLOCK), 0, st, segments,
                       num_segments, count, reduced, identity);
  if (count <= 0) return;
  const long long tiles = (count + LBS_TILE - 1) / LBS_TILE;
  if (segreduce_scratch_bytes(count, sizeof(T)) > ctx.scratch_bytes)
    throw mgx_error(MGX_E_INVALID, "segreduce: scratch arena too small");
  ++ctx.scratch_epoch;
  long long* carry_seg = (long long*)ctx.scratch;
  T* carry_val = (T*)(carry_seg + tiles * 2);
  MGX_HIP(hipMemsetAsync(carry_seg, 0xFF, (size_t)tiles * 2 * sizeof(long long), st));
  // (the first generation, k_lbs_segreduce -- runs folded serially out of LDS, 1.75 against 1.40 ms on RMAT-22 -- stayed selectable
  //  through MGX_SEGREDUCE_GEN=1 until round 6)
    hipLaunchKernelGGL((k_lbs_segreduce2<T, F, Op>), dim3((unsigned)tiles), dim3(BLOCK), 0, st, f, count, segments,
                       num_segments, reduced, op, identity, carry_val, carry_seg);
  hipLaunchKernelGGL((k_segreduce_fixup<T, Op>), dim3(grid_for(tiles)), dim3(BLOCK), 0, st, tiles, carry_val, carry_seg,
                     reduced, op);
}

}  // namespace mgx
