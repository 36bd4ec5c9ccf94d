// mgx/wave.hpp -- wave64 / workgroup building blocks for gfx950 (CDNA4).
// Everything here assumes 64-lane wavefronts and 256-thread workgroups (4 waves, one per SIMD).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace mgx {

constexpr int WAVE = 64;
constexpr int BLOCK = 256;
constexpr int WAVES_PER_BLOCK = BLOCK / WAVE;

typedef unsigned long long u64;
typedef unsigned int u32;

__device__ __forceinline__ int lane_id() {
  return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}
// number of set bits of `mask` strictly below the calling lane (v_mbcnt pair)
__device__ __forceinline__ int rank_in_mask(u64 mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((u32)(mask >> 32), __builtin_amdgcn_mbcnt_lo((u32)mask, 0u));
}
// XCD (accelerator complex die) this wave runs on, 0..7: s_getreg_b32 hwreg(HW_REG_XCC_ID, 0, 4).
// Used for placement-dependent SPEED only, never correctness.
__device__ __forceinline__ int xcc_id() {
  return (int)__builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 0xF;
}

template <typename T>
__device__ __forceinline__ T wave_inclusive_sum(T x) {
  const int lane = lane_id();
#pragma unroll
  for (int d = 1; d < WAVE; d <<= 1) {
    T y = __shfl_up(x, d, WAVE);
    if (lane >= d) x += y;
  }
  return x;
}
template <typename T>
__device__ __forceinline__ T wave_sum(T x) {
#pragma unroll
  for (int d = WAVE / 2; d > 0; d >>= 1) x += __shfl_xor(x, d, WAVE);
  return x;
}

// Exclusive sum over the BLOCK threads of a workgroup.  `smem` needs WAVES_PER_BLOCK+1 slots of T.
// Returns the exclusive prefix of x; *total gets the block sum.  Contains two barriers.
template <typename T>
__device__ __forceinline__ T block_exclusive_sum(T x, T* smem, T* total) {
  const int lane = lane_id();
  const int wave = threadIdx.x / WAVE;
  T inc = wave_inclusive_sum(x);
  if (lane == WAVE - 1) smem[wave] = inc;
  __syncthreads();
  T base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < WAVES_PER_BLOCK; ++w) {
    T s = smem[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - x;
}

// Lanes of ONE wave exchanging data through LDS without a workgroup barrier: the hardware executes a wave's LDS
// instructions in order, but the COMPILER reasons per thread -- a store to p[lane] and a load from p[lane + 1] do
// not alias for it, and it may hoist the load above the store (seen: k_sssp_relax read stale row offsets and
// faulted).  Call this between the writes and the reads of other lanes' slots.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Intra-wave duplicate cull (the reference's UniquifyFunctor::warp_cull, filter.hxx:58-76, for wave64): returns true
// for every calling lane but those that hold the same `item` as another calling lane of the wave which won their shared
// hash slot -- of k lanes with one item, at least one survives (exactly one unless a different item takes the slot:
// then all k survive; the cull is an optimisation in front of an exact test, never the test itself).  Lanes may call it
// under divergence (the tail of an array); workgroups of at most 16 waves.  1 LDS write, 1 LDS read, 1 shuffle.
__device__ __forceinline__ bool wave_first_of_equal(int item) {
  __shared__ int slots[16][256];
  const int lane = lane_id();
  int* const mine = slots[(threadIdx.x / WAVE) & 15];
  const unsigned h = ((unsigned)item * 2654435761u) >> 24;     // 256 slots
  mine[h] = lane;
  wave_lds_fence();
  const int winner = mine[h];                                   // a lane that wrote in THIS call (mine or a later one)
  const int theirs = __shfl(item, winner, WAVE);
  wave_lds_fence();                                             // (the slots are rewritten by the next call)
  return winner == lane || theirs != item;
}

// Same for a workgroup of NW waves (any block size); `smem` needs NW slots of T.  Two barriers.
template <int NW, typename T>
__device__ __forceinline__ T block_exclusive_sum_nw(T x, T* smem, T* total) {
  const int lane = lane_id();
  const int wave = threadIdx.x / WAVE;
  T inc = wave_inclusive_sum(x);
  if (lane == WAVE - 1) smem[wave] = inc;
  __syncthreads();
  T base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    T s = smem[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - x;
}

// The same with the wave totals scanned by wave 0's lanes instead of every thread reading all NW of them: the loop above
// reads NW uniform LDS words, which hipcc moves to scalar registers (2 NW SGPRs per call for 64-bit sums; several calls
// in flight spilled the chain of small levels).  `smem` needs NW + 1 slots.  Three barriers.
template <int NW, typename T>
__device__ __forceinline__ T block_exclusive_sum_lean(T x, T* smem, T* total) {
  static_assert(NW <= WAVE, "one lane per wave total");
  const int lane = lane_id();
  const int wave = threadIdx.x / WAVE;
  const T inc = wave_inclusive_sum(x);
  if (lane == WAVE - 1) smem[wave] = inc;
  __syncthreads();
  if (wave == 0) {
    const T v = lane < NW ? smem[lane] : T(0);
    const T s = wave_inclusive_sum(v);
    if (lane < NW) smem[lane] = s - v;
    if (lane == NW - 1) smem[NW] = s;
  }
  __syncthreads();
  const T base = smem[wave];
  *total = smem[NW];
  __syncthreads();
  return base + inc - x;
}

// Wave-cooperative upper bound on a sorted global array: returns the number of elements
// a[i] <= key for i in [0,n) (so the last position with a[pos] <= key is result-1).
// 64-ary search: every step narrows the window by 64x with one coalescable probe per lane.
// All lanes of the wave must call it with identical arguments; all lanes get the result.
template <typename T>
__device__ __forceinline__ long long wave_upper_bound(const T* __restrict__ a, long long n, T key) {
  const int lane = lane_id();
  long long lo = 0, hi = n;   // invariant: a[i] <= key for i < lo, a[i] > key for i >= hi
  while (hi - lo > WAVE) {
    const long long step = (hi - lo + WAVE - 1) / WAVE;     // >= 2
    const long long probe = lo + (long long)(lane + 1) * step - 1;   // last element of my slice
    bool le = (probe < hi) ? (a[probe] <= key) : false;
    const u64 m = __ballot(le);
    const int k = __popcll(m);             // slices fully <= key (monotone, so a prefix)
    const long long nlo = lo + (long long)k * step;
    const long long nhi = (nlo + step < hi) ? nlo + step : hi;
    lo = nlo < hi ? nlo : hi;
    hi = nhi;
  }
  {
    const long long probe = lo + lane;
    bool le = (probe < hi) ? (a[probe] <= key) : false;
    lo += __popcll(__ballot(le));
  }
  return lo;
}

}  // namespace mgx
