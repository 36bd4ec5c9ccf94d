// mgx/nreduce.hpp -- segmented neighbour-reduce over the FULL frontier on the graph's hub-first layout.
//
// The operator (gunrock/neighborhood.hxx, reference neighborhood.hxx:12-70) computes reduced[i] = (+) over the neighbours u
// of frontier[i] of value(u).  Its general kernel (lbs.hpp: k_lbs_segreduce2) addresses the edges by rank: a load-balanced
// search per item and one 4-byte gather of value(u) per edge out of an n-float array in arbitrary order -- 77 G gathers/s
// from 16 MB on this part (profiles/r01/microbench.jsonl), 1.4 ms for the 134 M edges of RMAT-22, 0.14 of the HBM roof.
// When the frontier is every vertex in order (PR's first iteration, mgx_segreduce_* over an iota frontier) and the graph
// carries the hub-first layout of the fused BFS, nothing needs a search and most gathers never leave the compute unit:
//
//   k_nr_values   (also checks that the frontier is 0 .. n - 1) vals[v] = value(old_of_new[v]) for every layout vertex v: ONE gather per vertex instead of one per
//                 edge; from here on the values are addressed by layout id, where the hubs -- the targets of most edges --
//                 are the first ids; reduced[] <- identity.
//   k_nr_edges    ONE launch, two parts.  Long rows (>= 64 entries) from the unit blocks (mgx_layout.hip): 16 bytes per lane, a unit of 64
//                 entries belongs to ONE row (its padding entries are -1: the identity), so the segment id is free;
//                 the values of the first NR_HOTV layout vertices sit in LDS (160 KB: one workgroup per CU), the others
//                 are gathered from L2 (the next 600 K vertices are 2.4 MB); 16 lanes fold a unit with
//                 four shuffle steps in a fixed order -> partial[u].
//   k_nr_fold     a long row's units are contiguous (ub_first[v] .. ub_first[v + 1]): one thread folds the partials of a
//                 row of up to NR_BIG_UNITS units (four accumulators, a fixed order); the few rows above that (the first rows
//                 of the degree-sorted layout) take a workgroup each.  Deterministic: the order of the fold never depends on timing.
//                 Short rows: 1 .. 63 entries by degree class as in bfs_fused_vshort.hpp: 16 / 4 / 1 lanes per vertex, one
//                 unaligned 16-byte load of four entries per lane, fold by shuffles.
// Results go to reduced[old_of_new[v]] -- the frontier POSITION of vertex v in an iota frontier (neighborhood.hxx:58).
// Float sums are folded in a different order than the general kernel's (both are deterministic; the reference's own
// order is moderngpu's and unpinned): the tests compare with 2e-5 relative.
#pragma once
#include <cstdlib>
#include <type_traits>

#include "runtime.hpp"
#include "wave.hpp"

namespace mgx {

constexpr int NR_HOTV = 40000;            // values of the first NR_HOTV layout vertices in LDS: 160 000 bytes, ONE workgroup per CU.
                                          // (Measured, RMAT-22: 0.539 ms per reduce against 0.609 with two workgroups of 20 000 values each --
                                          //  unlike the BFS's bit probes, every entry here is a 4-byte gather: what counts is how many of them
                                          //  stay in LDS -- 64 % of the endpoints at 40 000 values against 51 % -- and 128 registers per lane.)
constexpr int NR_BIG_UNITS = 64;          // rows of more units than this are folded by a workgroup of their own
constexpr size_t nr_lds_bytes() { return (size_t)NR_HOTV * 4 + 64; }
constexpr int NRS_MAX_SLICES = 96;        // hot slices of the sliced long rows (k_nrs_edges) at most
// How many a graph of n vertices gets (nrs_default_slices): 16 x 40 000 vertices hold 94 % of RMAT-22's long-row endpoints, but 82 / 63 % of
// RMAT-24's / 25's -- the tail behind the hot slices gathers through the L2 -- and every slice costs the fold a range end per long row.
// Measured (profiles/r06/nr_slices_sweep.txt, ms per call at 16 / 32 / 64 / 96-128 slices): RMAT-22 0.269 / 0.276 / 0.321 / --,
// RMAT-23 0.705 / 0.644 / 0.621-0.634 / 0.665, RMAT-24 1.917 / 1.594 / 1.438 / 1.449, RMAT-25 5.021 / 4.319 / 3.715 / 3.526.
inline int nrs_default_slices(long long n) {
  const long long by_size = n >> 18;                 // 16 / 32 / 64 / 128 at R-MAT 22 / 23 / 24 / 25
  return (int)(by_size < 16 ? 16 : by_size > NRS_MAX_SLICES ? NRS_MAX_SLICES : by_size);
}
constexpr int NRS_FOLD_CHUNK = 17;        // slices (hot + tail) k_nrs_fold takes at a time: a row's range ends of that many sit in registers
// k_nrs_fold: rows of more than NRS_FOLD_DEG[0] entries are folded by a workgroup each, of more than [1] by a wave, of more than [2] by
// eight lanes, the others by a thread each (the layout is sorted by degree: the tiers are row ranges)
constexpr int NRS_FOLD_DEG[3] = {65536, 4096, 256};

struct nr_layout_t {
  const u32* row_offsets = nullptr;       // the layout's CSR
  const int* col_indices = nullptr;
  const int* old_of_new = nullptr;
  const int* ub_col = nullptr;            // unit blocks of the rows of >= 64 entries
  const u32* ub_col24 = nullptr;          // the same entries, 24 bits each (graphs of at most 2^23 vertices; NULL: none): 12 instead of 16 bytes per lane and load
  const unsigned char* ub_cnt = nullptr;  // real entries of every unit (1 .. 64; 0 for padding units)
  const int* ub_first = nullptr;          // n + 1: units of row v = [ub_first[v], ub_first[v + 1])
  u32 ub_units = 0, ub_units_pad = 0;
  u32 vs_v[4] = {0, 0, 0, 0};             // degree classes of the short rows (bfs_fused_vshort.hpp)
  u32 vs_dummy = 0;                       // index into col_indices of four entries of -1
  u32 big_rows = 0;                       // rows [0, big_rows) hold more than NR_BIG_UNITS units each
  int n = 0;
  // the long rows by slice of their destinations (round 5, mgx_layout.hip: mgx_nrs_build_device; nrs_mu == NULL: not built):
  // 16-byte mini-units, slice-major -- slices [0, nrs_slices) as 8 x 16-bit offsets into the slice (padding: NR_HOTV), slice
  // nrs_slices (the tail: everything behind nrs_slices * NR_HOTV) as 4 x 32-bit layout ids (padding: -1)
  const uint4* nrs_mu = nullptr;
  const u32* nrs_off = nullptr;           // (nrs_slices + 1) * nrs_rows + 1: the mini-units of (slice k, row r) start at nrs_off[k * nrs_rows + r]
  u32 nrs_first[NRS_MAX_SLICES + 2] = {}; // first mini-unit of slice k; [nrs_slices + 1] = all of them
  u32 nrs_slices = 0;
  u32 nrs_rows = 0;                       // the long rows: [0, nrs_rows) (= vs_v[0])
  u32 nrs_tier[3] = {0, 0, 0};            // the fold's tiers: rows [0, t0) a workgroup each, [t0, t1) a wave, [t1, t2) eight lanes, [t2, nrs_rows) a thread
  u32 parts = 3u;                         // (timing runs, MGX_NR_PARTS: 1 the short rows only, 2 the long rows only -- the results are then incomplete)
  // a frontier that is a SUBSET of the vertices (round 6; pos == NULL: the full frontier 0 .. n - 1, results by original id): layout
  // vertex v is at frontier position (u32)pos[v] iff pos[v] >> 32 == pos_epoch (k_nr_values_subset wrote it in THIS call: the
  // array is never cleared); results go to reduced[position] (neighborhood.hxx:58), rows outside the frontier are computed and dropped
  const int* new_of_old = nullptr;
  const u64* pos = nullptr;
  u32 pos_epoch = 0;
};

// where the result of layout vertex v goes
template <typename V>
__device__ __forceinline__ void nr_store(const nr_layout_t& L, V* __restrict__ reduced, u32 v, V x) {
  if (!L.pos) { reduced[L.old_of_new[v]] = x; return; }
  const u64 p = L.pos[v];
  if ((u32)(p >> 32) == L.pos_epoch) reduced[(u32)p] = x;
}

// k_nr_values also answers: is the frontier 0, 1, ..., n - 1?  *host_flag (pinned) was set to 1 by the host before the launch;
// *dev_flag <- epoch if it is not: the kernels behind this one return at once when they find their epoch there (no host wait
// between the check and the work; the host looks at host_flag when everything has run and takes the general kernel if the
// answer was no -- vals[] is scratch and reduced[] is rewritten by that kernel, so what this one stored does no harm).
// Epochs only grow: nothing is ever reset.
template <typename V, typename GetValue>
__global__ __launch_bounds__(BLOCK) void k_nr_values(GetValue get, const int* __restrict__ old_of_new, V* __restrict__ vals,
                                                     V* __restrict__ reduced, V identity, long long n, const int* __restrict__ frontier,
                                                     long long* host_flag, u32* dev_flag, u32 epoch) {
  bool bad = false;
  for (long long v = (long long)blockIdx.x * BLOCK + threadIdx.x; v < n; v += (long long)gridDim.x * BLOCK) {
    bad |= frontier[v] != (int)v;
    vals[v] = get(old_of_new[v]);
    reduced[v] = identity;
  }
  if (__ballot(bad) && lane_id() == 0) { *host_flag = 0; *dev_flag = epoch; }
}

// The same for a frontier that is a subset: f[0] < f[1] < ... < f[nf - 1], all in [0, n) -- what a stable filter leaves of an iota
// (pr_enactor.hxx: every iteration but the first) -- checked here the same way (*host_flag <- 0, *dev_flag <- epoch otherwise: a
// frontier with duplicates or out of order takes the general kernel).  vals[] as above for ALL vertices (any of them may be a
// neighbour); pos[new_of_old[f[i]]] = (epoch, i); reduced[i] <- identity for the nf positions; the frontier's degrees are summed into
// *edges (the operator's return value: monotonic, the host keeps the base).
template <typename V, typename GetValue>
__global__ __launch_bounds__(BLOCK) void k_nr_values_subset(GetValue get, const int* __restrict__ old_of_new, const int* __restrict__ new_of_old,
                                                            V* __restrict__ vals, V* __restrict__ reduced, V identity, long long n,
                                                            const int* __restrict__ frontier, long long nf, const int* __restrict__ offsets,
                                                            u64* __restrict__ pos, u64* edges, long long* host_flag, u32* dev_flag, u32 epoch) {
  const long long t0 = (long long)blockIdx.x * BLOCK + threadIdx.x, nt = (long long)gridDim.x * BLOCK;
  for (long long v = t0; v < n; v += nt) vals[v] = get(old_of_new[v]);
  bool bad = false;
  u64 deg = 0;
  for (long long i = t0; i < nf; i += nt) {
    const int f = frontier[i];
    const int prev = i ? frontier[i - 1] : -1;
    const bool ok = f >= 0 && f > prev && (long long)f < n;
    bad |= !ok;
    if (ok) {
      pos[new_of_old[f]] = ((u64)epoch << 32) | (u64)(u32)i;
      deg += (u64)(u32)(offsets[f + 1] - offsets[f]);
    }
    reduced[i] = identity;
  }
  if (__ballot(bad) && lane_id() == 0) { *host_flag = 0; *dev_flag = epoch; }
  // ONE add per workgroup (an add per wave -- 37 000 of them on one address for RMAT-22's 2.4 M frontier vertices -- made this kernel
  // 130 us instead of 45: same-address atomics are served one after the other, ~5-11 ns each, tools/microbench5.hip)
  __shared__ unsigned long long s_deg;
  if (threadIdx.x == 0) s_deg = 0ull;
  __syncthreads();
  deg = wave_sum(deg);
  if (lane_id() == 0 && deg) atomicAdd(&s_deg, (unsigned long long)deg);
  __syncthreads();
  if (threadIdx.x == 0 && s_deg) atomicAdd((unsigned long long*)edges, s_deg);
}
// last kernel of a call: (subset calls) the degree sum to the host's mailbox, then the sequence number the host spins on
// (standard_context_t::mailbox_wait: a hipStreamSynchronize wakes up 10-20 us later)
__global__ void k_nr_publish(const u64* edges, long long* host_edges, long long* mailbox, long long seq) {
  if (host_edges) *host_edges = (long long)*edges;
  __threadfence_system();
  __hip_atomic_store(mailbox + 1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// value of entry d: -1 (padding, lanes past a row's end) -> identity, a hub -> LDS, anything else -> L2 / HBM.  The global
// load is unconditional (a load under a condition serialises the pipeline, bfs_fused.hpp "countable loads"): entries
// served from LDS read vals[0] instead, one broadcast request per wave instruction.
// A load the compiler must leave where it stands.  A plain load whose only use is one arm of a select is SUNK under the select's
// condition by the code generator (select -> branch around the load), and hipcc's s_waitcnt insertion then waits for vmcnt(0)
// inside every such branch: every gather of a step became a full round trip of its own, with the prefetched entry loads drained
// on top (found in the ISA in round 5: 70 "s_waitcnt vmcnt(0) lgkmcnt(0)" in k_nr_edges).  A relaxed atomic load of wavefront
// scope is the same machine instruction without cache-policy bits, but not a candidate for that transformation.
template <typename V>
__device__ __forceinline__ V nr_load_pinned(const V* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
}

template <typename V>
__device__ __forceinline__ V nr_fetch(u32 d, const V* __restrict__ vals, const V* hot, u32 hot_n, V identity) {
  const bool is_hot = d < hot_n;
  const bool none = (int)d < 0;
  const V g = nr_load_pinned(vals + ((is_hot || none) ? 0u : d));
  const V h = nr_load_pinned(hot + (is_hot ? d : 0u));
  return none ? identity : (is_hot ? h : g);
}

template <typename V, int NT>
__device__ __forceinline__ V* nr_hot_setup(char* smem, const V* __restrict__ vals, u32 hot_n) {
  V* const hot = (V*)smem;
  for (u32 i = threadIdx.x; i < hot_n; i += NT) hot[i] = vals[i];
  __syncthreads();
  return hot;
}

struct __attribute__((aligned(4))) nr_u32x4u { u32 x, y, z, w; };   // 16-byte load at 4-byte alignment
typedef unsigned int nr_u32x4 __attribute__((ext_vector_type(4)));

// the long rows' part of one workgroup (block `block` of `nblocks`); hot: the LDS values, already set up
typedef unsigned int nr_u32x3 __attribute__((ext_vector_type(3)));
template <typename V, typename Op, int NT, bool P24 = false>
__device__ __forceinline__ void nr_long_work(const nr_layout_t& L, const V* __restrict__ vals, const V* hot, u32 hot_n, V* __restrict__ partial,
                                             V identity, Op op, u32 block, u32 nblocks) {
  constexpr int NW = NT / WAVE;
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
  const u32 G = L.ub_units_pad / 16u;                 // groups of 16 units = 1024 entries
  const u32 W = nblocks * NW, w = block * NW + (u32)wave;
  const int* __restrict__ ucol = L.ub_col;
  const u32 sub = (u32)lane & 15u, q = (u32)lane >> 4;                  // my four entries inside my unit; my unit inside a 4-unit load
  if (w < G) {
  // A step takes GPS groups (g, g + W, ...): four loads per group; load j covers units 16 g + 4 j .. + 3, lane (q, sub) reads
  // entries 4 sub .. 4 sub + 3 of unit 4 j + q.  The next step's loads are in flight while this step's 16 GPS gathers are.
  constexpr int GPS = 1, NL = 4 * GPS;      // (two groups per step until round 5: with the gathers pinned (nr_load_pinned) all 32 + 32 of a step are in flight at once and spill)
  // (a unit's padding entries are -1 -- the identity to nr_fetch -- so nothing has to say how many of its 64 entries are real:
  //  the per-unit counts this loop used to load, one byte per lane and 16-byte load, doubled its memory instructions)
  typedef typename std::conditional<P24, nr_u32x3, nr_u32x4>::type raw_t;
  const u32* __restrict__ ucol24 = L.ub_col24;
  raw_t cur[NL], nxt[NL];
  auto issue = [&](u32 g0, raw_t* d) {
#pragma unroll
    for (int k = 0; k < GPS; ++k) {
      const u32 g = g0 + (u32)k * W;
      const u32 gg = g < G ? g : G - 1u;                // (past the end: the last group again, masked below)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const u32 u = gg * 16u + 4u * (u32)j + q;
        if constexpr (P24) d[4 * k + j] = __builtin_nontemporal_load((const nr_u32x3*)(ucol24 + (((size_t)u << 4) + sub) * 3u));
        else d[4 * k + j] = __builtin_nontemporal_load((const nr_u32x4*)(ucol + ((size_t)u << 6) + sub * 4u));
      }
    }
  };
  issue(w, cur);
  for (u32 g0 = w; g0 < G; g0 += (u32)GPS * W) {
    issue(g0 + (u32)GPS * W, nxt);
    // all gathers of the step first, then the folds (a fold's shuffles between two gathers would order them)
    V val[NL][4];
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      const bool real = g0 + (u32)(j / 4) * W < G;    // (wave-uniform)
      u32 e0, e1, e2, e3;
      if constexpr (P24) {                            // (sign-extending: 0xFFFFFF comes back as -1)
        e0 = (u32)__builtin_amdgcn_sbfe((int)cur[j].x, 0, 24);
        e1 = (u32)__builtin_amdgcn_sbfe((int)__builtin_amdgcn_alignbit(cur[j].y, cur[j].x, 24), 0, 24);
        e2 = (u32)__builtin_amdgcn_sbfe((int)__builtin_amdgcn_alignbit(cur[j].z, cur[j].y, 16), 0, 24);
        e3 = (u32)((int)cur[j].z >> 8);
      } else { e0 = cur[j].x; e1 = cur[j].y; e2 = cur[j].z; e3 = cur[j].w; }
      const u32 d0 = real ? e0 : 0xFFFFFFFFu, d1 = real ? e1 : 0xFFFFFFFFu;
      const u32 d2 = real ? e2 : 0xFFFFFFFFu, d3 = real ? e3 : 0xFFFFFFFFu;
      val[j][0] = nr_fetch(d0, vals, hot, hot_n, identity); val[j][1] = nr_fetch(d1, vals, hot, hot_n, identity);
      val[j][2] = nr_fetch(d2, vals, hot, hot_n, identity); val[j][3] = nr_fetch(d3, vals, hot, hot_n, identity);
    }
#pragma unroll
    for (int j = 0; j < NL; ++j) {
      V s = op(op(val[j][0], val[j][1]), op(val[j][2], val[j][3]));
#pragma unroll
      for (int sh = 1; sh < 16; sh <<= 1) s = op(s, __shfl_xor(s, sh, WAVE));     // the 16 lanes of my unit
      const u32 g = g0 + (u32)(j / 4) * W;
      if (sub == 0u && g < G) partial[g * 16u + 4u * (u32)(j % 4) + q] = s;
    }
#pragma unroll
    for (int j = 0; j < NL; ++j) cur[j] = nxt[j];
  }
  }
}

// A long row's partials -> its result, ONE launch: workgroups [0, big_rows) take a row of more than NR_BIG_UNITS units each (a
// fixed strided fold), the others one row per thread of [big_rows, last_row): four accumulators over the units k % 4 (four
// independent loads in flight instead of a chain of up to 64 dependent ones), combined in a fixed order.
template <typename V, typename Op>
__global__ __launch_bounds__(BLOCK) void k_nr_fold(nr_layout_t L, const V* __restrict__ partial, V* __restrict__ reduced, V identity, Op op,
                                                   u32 last_row, const u32* dev_flag, u32 epoch) {
  __shared__ V s_part[BLOCK / WAVE];
  if (*dev_flag == epoch) return;
  if (blockIdx.x < L.big_rows) {
    const u32 r = blockIdx.x;
    const int u0 = L.ub_first[r], u1 = L.ub_first[r + 1];
    V acc = identity;
    for (int u = u0 + (int)threadIdx.x; u < u1; u += BLOCK) acc = op(acc, partial[u]);
#pragma unroll
    for (int sh = 1; sh < WAVE; sh <<= 1) acc = op(acc, __shfl_xor(acc, sh, WAVE));
    if (lane_id() == 0) s_part[threadIdx.x / WAVE] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      V t = s_part[0];
#pragma unroll
      for (int k = 1; k < BLOCK / WAVE; ++k) t = op(t, s_part[k]);
      nr_store(L, reduced, r, t);
    }
    return;
  }
  const u32 r = L.big_rows + (blockIdx.x - L.big_rows) * BLOCK + threadIdx.x;
  if (r >= last_row) return;
  const int u0 = L.ub_first[r], u1 = L.ub_first[r + 1];
  if (u1 <= u0) return;
  V a0 = identity, a1 = identity, a2 = identity, a3 = identity;
  int u = u0;
  for (; u + 4 <= u1; u += 4) {
    const V p0 = partial[u], p1 = partial[u + 1], p2 = partial[u + 2], p3 = partial[u + 3];
    a0 = op(a0, p0); a1 = op(a1, p1); a2 = op(a2, p2); a3 = op(a3, p3);
  }
  if (u < u1) a0 = op(a0, partial[u]);
  if (u + 1 < u1) a1 = op(a1, partial[u + 1]);
  if (u + 2 < u1) a2 = op(a2, partial[u + 2]);
  nr_store(L, reduced, r, op(op(a0, a1), op(a2, a3)));
}

// short rows by degree class: [vs_v[0], vs_v[1]) 16 lanes per vertex (17 .. 63 entries), [vs_v[1], vs_v[2]) 4 lanes (5 .. 16),
// [vs_v[2], vs_v[3]) 1 lane (1 .. 4).  A lane reads four consecutive entries of its row with one 16-byte load.
template <typename V, typename Op, int NT>
__device__ __forceinline__ void nr_short_work(const nr_layout_t& L, const V* __restrict__ vals, const V* hot, u32 hot_n, V* __restrict__ reduced,
                                              V identity, Op op, u32 block, u32 nblocks) {
  constexpr int NW = NT / WAVE;
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
  const u32 b0 = L.vs_v[0], b1 = L.vs_v[1], b2 = L.vs_v[2], b3 = L.vs_v[3];
  const u32 s16 = (b1 - b0 + 3u) / 4u, s4 = (b2 - b1 + 15u) / 16u, s1 = (b3 - b2 + 63u) / 64u;
  const u32 T = s16 + s4 + s1;
  const u32 W = nblocks * NW, w = block * NW + (u32)wave;
  const u32* __restrict__ ro = L.row_offsets;
  const int* __restrict__ col = L.col_indices;
  struct plan_t { u32 e0, cnt, v, lpr_shift; };
  auto plan = [&](u32 s) -> plan_t {
    plan_t p; p.e0 = L.vs_dummy; p.cnt = 0; p.v = 0xFFFFFFFFu; p.lpr_shift = 0;
    u32 vbase, vend;
    if (s < s16) { p.lpr_shift = 4; vbase = b0 + s * 4u; vend = b1; }
    else if (s < s16 + s4) { p.lpr_shift = 2; vbase = b1 + (s - s16) * 16u; vend = b2; }
    else { p.lpr_shift = 0; vbase = b2 + (s - s16 - s4) * 64u; vend = b3; }
    const u32 v = vbase + ((u32)lane >> p.lpr_shift);
    const u32 sub = (u32)lane & ((1u << p.lpr_shift) - 1u);
    const bool in = s < T && v < vend;
    const u32 vc = in ? v : 0u;
    const u32 lo = ro[vc], hi = ro[vc + 1];
    const u32 deg = hi - lo;
    if (in) p.v = v;
    if (in && sub * 4u < deg) { p.e0 = lo + sub * 4u; p.cnt = deg - sub * 4u < 4u ? deg - sub * 4u : 4u; }
    return p;
  };
  plan_t pc = plan(w);
  nr_u32x4u dc = *(const nr_u32x4u*)(col + pc.e0);
  for (u32 s = w; s < T; s += W) {
    const plan_t pn = plan(s + W);
    const nr_u32x4u dn = *(const nr_u32x4u*)(col + pn.e0);
    const u32 d0 = pc.cnt > 0u ? dc.x : 0xFFFFFFFFu, d1 = pc.cnt > 1u ? dc.y : 0xFFFFFFFFu;
    const u32 d2 = pc.cnt > 2u ? dc.z : 0xFFFFFFFFu, d3 = pc.cnt > 3u ? dc.w : 0xFFFFFFFFu;
    const V v0 = nr_fetch(d0, vals, hot, hot_n, identity), v1 = nr_fetch(d1, vals, hot, hot_n, identity);
    const V v2 = nr_fetch(d2, vals, hot, hot_n, identity), v3 = nr_fetch(d3, vals, hot, hot_n, identity);
    V acc = op(op(v0, v1), op(v2, v3));
    // fold over the lanes of my vertex (wave-uniform class: the step decides it)
    if (pc.lpr_shift == 4) {
#pragma unroll
      for (int sh = 1; sh < 16; sh <<= 1) acc = op(acc, __shfl_xor(acc, sh, WAVE));
    } else if (pc.lpr_shift == 2) {
#pragma unroll
      for (int sh = 1; sh < 4; sh <<= 1) acc = op(acc, __shfl_xor(acc, sh, WAVE));
    }
    const u32 sub = (u32)lane & ((1u << pc.lpr_shift) - 1u);
    if (pc.v != 0xFFFFFFFFu && sub == 0u) nr_store(L, reduced, pc.v, acc);
    pc = pn; dc = dn;
  }
}

// ONE launch for both parts.
// HOTV values in LDS, WPE waves per SIMD: one workgroup of 1024 threads per CU
template <typename V, typename Op, int NT, int HOTV = NR_HOTV, int WPE = 4>
__global__ __launch_bounds__(NT, WPE) void k_nr_edges(nr_layout_t L, const V* __restrict__ vals, V* __restrict__ partial, V* __restrict__ reduced,
                                                    V identity, Op op, const u32* dev_flag, u32 epoch) {
  static_assert(sizeof(V) == 4, "k_nr_edges keeps HOTV 4-byte values in LDS (nr_lds_bytes): size both by sizeof(V) before adding an 8-byte value type");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (*dev_flag == epoch) return;                    // (grid-uniform)
  const u32 hot_n = (u32)L.n < (u32)HOTV ? (u32)L.n : (u32)HOTV;
  const V* const hot = nr_hot_setup<V, NT>(smem, vals, hot_n);
  // every workgroup takes its share of BOTH parts, one after the other over the same LDS values: the parts differ in cost
  // per entry (the short rows pay a planning load per vertex), so any fixed split of the grid leaves one half waiting
  if (L.parts & 2u) {
    if (L.ub_col24) nr_long_work<V, Op, NT, true>(L, vals, hot, hot_n, partial, identity, op, blockIdx.x, gridDim.x);      // (grid-uniform)
    else nr_long_work<V, Op, NT, false>(L, vals, hot, hot_n, partial, identity, op, blockIdx.x, gridDim.x);
  }
  if (L.parts & 1u) nr_short_work<V, Op, NT>(L, vals, hot, hot_n, reduced, identity, op, blockIdx.x, gridDim.x);
}


// ---- the long rows by slice of their destinations (round 5) -------------------------------------------------------------
// k_nr_edges keeps the values of the first NR_HOTV layout vertices in LDS; every other entry is a 4-byte gather through the L2
// -- 40 % of RMAT-22's long-row entries, 0.9 GB of sector traffic for 366 MB of entries (profiles/r05/pmc_by_kernel.txt):
// what bounded the operator.  With the entries regrouped by SLICE of their destination (nr_layout_t::nrs_*) a workgroup takes
// mini-units of one slice at a time, that slice's values in LDS: a lane loads 16 bytes -- eight 16-bit offsets --, reads eight
// values from LDS, folds them in a fixed order and stores ONE partial; no gather leaves the compute unit, no shuffle, no
// owner is looked up (the fold kernel knows where a row's partials are).  Only the tail behind the last hot slice (6 % of the
// entries) still gathers.  The workgroups split the mini-unit sequence into equal contiguous shares: a share touches one or
// two slices (the first slice is 60 % of everything), so a workgroup loads one or two tables.  In front of its share every
// workgroup takes its part of the SHORT rows (nr_short_work) over the table of slice 0 -- the same values k_nr_edges keeps.
// Measured on RMAT-22 (profiles/r05/nr_sliced_ab.log; one operator call 0.487 -> 0.265 ms): k_nr_edges 408 us = long rows 336 + short rows 58;
// k_nrs_edges 164-168 us = mini-units 116 (17.5 M hot + 1.9 M tail) + short rows 53-57, k_nrs_fold 39-40, k_nr_values 36.  Tried and dropped
// in the same round: eight mini-units in flight per lane instead of four (116.0 against 116.1 us); a few waves of the workgroups that
// stay in slice 0 taking the short rows WHILE the others stream (2 / 4 / 6 / 8 waves: 607 / 360 / 297 / 249 us against 165 -- the short
// rows' time is inversely proportional to the waves that work on them); the short rows as a four-stage pipeline (extents, entries,
// values, fold one step apart each: 57.4 against 55.8 us -- not the exposed round trips either).
template <typename V, typename Op, int NT, int U>
__device__ __forceinline__ void nrs_hot_pass(const uint4* __restrict__ mu, const V* hot, V* __restrict__ partial, u32 lo, u32 hi, Op op) {
  nr_u32x4 cur[U], nxt[U];
  auto issue = [&](u32 j0, nr_u32x4* d) {
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const u32 jj = j0 + (u32)q * NT;
      d[q] = __builtin_nontemporal_load((const nr_u32x4*)mu + (jj < hi ? jj : hi - 1u));
    }
  };
  u32 j0 = lo + threadIdx.x;
  if (j0 >= hi) return;
  issue(j0, cur);
  for (; j0 < hi; j0 += (u32)U * NT) {
    issue(j0 + (u32)U * NT, nxt);
    V s[U];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const V v0 = hot[cur[q].x & 0xFFFFu], v1 = hot[cur[q].x >> 16], v2 = hot[cur[q].y & 0xFFFFu], v3 = hot[cur[q].y >> 16];
      const V v4 = hot[cur[q].z & 0xFFFFu], v5 = hot[cur[q].z >> 16], v6 = hot[cur[q].w & 0xFFFFu], v7 = hot[cur[q].w >> 16];
      s[q] = op(op(op(v0, v1), op(v2, v3)), op(op(v4, v5), op(v6, v7)));
    }
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const u32 jj = j0 + (u32)q * NT;
      if (jj < hi) partial[jj] = s[q];
    }
#pragma unroll
    for (int q = 0; q < U; ++q) cur[q] = nxt[q];
  }
}

// the tail: four layout ids per mini-unit, values gathered from the array (unconditional loads: padding reads vals[0])
template <typename V, typename Op, int NT>
__device__ __forceinline__ void nrs_tail_pass(const uint4* __restrict__ mu, const V* __restrict__ vals, V* __restrict__ partial, u32 lo, u32 hi,
                                              V identity, Op op) {
  constexpr int U = 4;
  nr_u32x4 cur[U], nxt[U];
  auto issue = [&](u32 j0, nr_u32x4* d) {
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const u32 jj = j0 + (u32)q * NT;
      d[q] = __builtin_nontemporal_load((const nr_u32x4*)mu + (jj < hi ? jj : hi - 1u));
    }
  };
  u32 j0 = lo + threadIdx.x;
  if (j0 >= hi) return;
  issue(j0, cur);
  for (; j0 < hi; j0 += (u32)U * NT) {
    issue(j0 + (u32)U * NT, nxt);
    V g[U][4];
#pragma unroll
    for (int q = 0; q < U; ++q) {
      g[q][0] = nr_load_pinned(vals + ((int)cur[q].x < 0 ? 0u : cur[q].x)); g[q][1] = nr_load_pinned(vals + ((int)cur[q].y < 0 ? 0u : cur[q].y));
      g[q][2] = nr_load_pinned(vals + ((int)cur[q].z < 0 ? 0u : cur[q].z)); g[q][3] = nr_load_pinned(vals + ((int)cur[q].w < 0 ? 0u : cur[q].w));
    }
#pragma unroll
    for (int q = 0; q < U; ++q) {
      const V v0 = (int)cur[q].x < 0 ? identity : g[q][0], v1 = (int)cur[q].y < 0 ? identity : g[q][1];
      const V v2 = (int)cur[q].z < 0 ? identity : g[q][2], v3 = (int)cur[q].w < 0 ? identity : g[q][3];
      const u32 jj = j0 + (u32)q * NT;
      if (jj < hi) partial[jj] = op(op(v0, v1), op(v2, v3));
    }
#pragma unroll
    for (int q = 0; q < U; ++q) cur[q] = nxt[q];
  }
}

template <typename V, typename Op, int NT, int U = 4, int WPE = 4>
__global__ __launch_bounds__(NT, WPE) void k_nrs_edges(nr_layout_t L, const V* __restrict__ vals, V* __restrict__ partial, V* __restrict__ reduced,
                                                     V identity, Op op, const u32* dev_flag, u32 epoch) {
  static_assert(sizeof(V) == 4, "k_nrs_edges keeps NR_HOTV 4-byte values in LDS (nr_lds_bytes)");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if (*dev_flag == epoch) return;                    // (grid-uniform)
  constexpr u32 S = (u32)NR_HOTV;
  const u32 n = (u32)L.n;
  const u32 hot_n = n < S ? n : S;
  if (threadIdx.x == 0) ((V*)smem)[S] = identity;    // what a padding offset reads (nr_lds_bytes has the room)
  V* const hot = nr_hot_setup<V, NT>(smem, vals, hot_n);
  if (L.parts & 1u) nr_short_work<V, Op, NT>(L, vals, hot, hot_n, reduced, identity, op, blockIdx.x, gridDim.x);
  if (!(L.parts & 2u)) return;
  // every workgroup takes an equal contiguous share of the HOT mini-units (one or two slices: one or two tables) and an equal share
  // of the TAIL's: a tail mini-unit costs four gathers through the compute unit's vector memory path, and with the tail at the end
  // of one sequence the last 25 workgroups did all 7.5 M of them (433 us for the long rows' part against 60 for everybody else)
  const u32 K = L.nrs_slices, H = L.nrs_first[K], M = L.nrs_first[K + 1];
  const u32 lo_b = (u32)((u64)H * blockIdx.x / gridDim.x), hi_b = (u32)((u64)H * (blockIdx.x + 1u) / gridDim.x);
  u32 loaded = 0u;                                   // the slice whose values are in LDS
  for (u32 k = 0; k < K; ++k) {
    const u32 f0 = L.nrs_first[k], f1 = L.nrs_first[k + 1];
    const u32 lo = lo_b > f0 ? lo_b : f0, hi = hi_b < f1 ? hi_b : f1;
    if (lo >= hi) continue;                          // (workgroup-uniform)
    if (k != loaded) {
      __syncthreads();                               // (every wave is done with the table in LDS)
      const u32 base = k * S, cnt = n - base < S ? n - base : S;
      for (u32 i = threadIdx.x; i < cnt; i += NT) hot[i] = vals[base + i];
      __syncthreads();
      loaded = k;
    }
    nrs_hot_pass<V, Op, NT, U>(L.nrs_mu, hot, partial, lo, hi, op);
  }
  if (M > H) {
    const u32 T = M - H;
    const u32 lo = H + (u32)((u64)T * blockIdx.x / gridDim.x), hi = H + (u32)((u64)T * (blockIdx.x + 1u) / gridDim.x);
    if (lo < hi) nrs_tail_pass<V, Op, NT>(L.nrs_mu, vals, partial, lo, hi, identity, op);
  }
}

// a long row's partials -> its result: the row's mini-units of slice k are [nrs_off[k * rows + r], nrs_off[k * rows + r + 1]), for
// neighbouring rows neighbouring pieces of `partial`.  LANES lanes fold ONE row: all of its ranges' ends first (independent loads),
// then every range's first LANES partials (again all in flight at once: for most rows that is everything), then what is left of
// the long ranges.  The first version walked the slices one after the other -- a dependent pair of round trips per slice, 17 in a
// row, a workgroup each for 16 515 rows: 93 us.  Fixed strides, fixed order: deterministic.
template <typename V, typename Op, int LANES, bool MULTI>
__device__ __forceinline__ V nrs_fold_row(const u32* __restrict__ off, const V* __restrict__ partial, u32 K1, u32 LR, u32 r, u32 lane, V identity, Op op) {
  // (round 6: the slices in chunks of NRS_FOLD_CHUNK -- a layout of up to 16 hot slices + the tail is one chunk, the fold it always was;
  //  bigger graphs carry more slices and go round again with the same registers: MULTI, a kernel of its own so that the one-chunk fold
  //  keeps its code)
  constexpr int KM = NRS_FOLD_CHUNK;
  V acc = identity;
  for (u32 kc = 0; kc < (MULTI ? K1 : 1u); kc += (u32)KM) {           // (uniform)
    u32 a[KM], b[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      const u32 kk = kc + (u32)k;
      const size_t at = (size_t)(kk < K1 ? kk : 0u) * LR + r;
      a[k] = nr_load_pinned(off + at);
      b[k] = nr_load_pinned(off + at + 1u);
    }
    V first[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      const u32 j = a[k] + lane;
      const bool ok = kc + (u32)k < K1 && j < b[k];
      first[k] = nr_load_pinned(partial + (ok ? j : 0u));
      if (!ok) first[k] = identity;
    }
#pragma unroll
    for (int k = 0; k < KM; ++k) acc = op(acc, first[k]);
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      if (kc + (u32)k < K1) {
        V a0 = identity, a1 = identity;
        u32 j = a[k] + lane + (u32)LANES;
        for (; j + (u32)LANES < b[k]; j += 2u * LANES) {
          const V p0 = partial[j], p1 = partial[j + (u32)LANES];
          a0 = op(a0, p0); a1 = op(a1, p1);
        }
        if (j < b[k]) a0 = op(a0, partial[j]);
        acc = op(acc, op(a0, a1));
      }
    }
  }
  return acc;
}

template <typename V, typename Op, bool MULTI>
__global__ __launch_bounds__(BLOCK) void k_nrs_fold(nr_layout_t L, const V* __restrict__ partial, V* __restrict__ reduced, V identity, Op op,
                                                    const u32* dev_flag, u32 epoch) {
  constexpr int NW = BLOCK / WAVE;
  __shared__ V s_part[NW];
  if (*dev_flag == epoch) return;
  const u32 K1 = L.nrs_slices + 1u, LR = L.nrs_rows;
  const u32* __restrict__ off = L.nrs_off;
  const u32 t0 = L.nrs_tier[0], t1 = L.nrs_tier[1], t2 = L.nrs_tier[2];          // t0 <= t1 <= t2 <= LR
  const u32 b1 = (t1 - t0 + NW - 1u) / NW, b2 = (t2 - t1 + BLOCK / 8 - 1u) / (BLOCK / 8);
  u32 blk = blockIdx.x;
  if (blk < t0) {                                                  // a workgroup per row
    const u32 r = blk;
    V acc = nrs_fold_row<V, Op, BLOCK, MULTI>(off, partial, K1, LR, r, threadIdx.x, identity, op);
#pragma unroll
    for (int sh = 1; sh < WAVE; sh <<= 1) acc = op(acc, __shfl_xor(acc, sh, WAVE));
    if (lane_id() == 0) s_part[threadIdx.x / WAVE] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
      V t = s_part[0];
#pragma unroll
      for (int k = 1; k < NW; ++k) t = op(t, s_part[k]);
      nr_store(L, reduced, r, t);
    }
    return;
  }
  blk -= t0;
  if (blk < b1) {                                                  // a wave per row
    const u32 r = t0 + blk * NW + (u32)__builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    if (r >= t1) return;
    V acc = nrs_fold_row<V, Op, WAVE, MULTI>(off, partial, K1, LR, r, (u32)lane_id(), identity, op);
#pragma unroll
    for (int sh = 1; sh < WAVE; sh <<= 1) acc = op(acc, __shfl_xor(acc, sh, WAVE));
    if (lane_id() == 0) nr_store(L, reduced, r, acc);
    return;
  }
  blk -= b1;
  if (blk < b2) {                                                  // eight lanes per row
    const u32 r = t1 + blk * (BLOCK / 8) + threadIdx.x / 8u, sub = threadIdx.x & 7u;
    const bool in = r < t2;
    V acc = nrs_fold_row<V, Op, 8, MULTI>(off, partial, K1, LR, in ? r : t1, sub, identity, op);
#pragma unroll
    for (int sh = 1; sh < 8; sh <<= 1) acc = op(acc, __shfl_xor(acc, sh, WAVE));
    if (in && sub == 0u) nr_store(L, reduced, r, acc);
    return;
  }
  blk -= b2;
  const u32 r = t2 + blk * BLOCK + threadIdx.x;                    // a thread per row
  if (r >= LR) return;
  nr_store(L, reduced, r, nrs_fold_row<V, Op, 1, MULTI>(off, partial, K1, LR, r, 0u, identity, op));
}

// scratch the fast path needs (vals + partials: one per unit, or per mini-unit of the sliced long rows), in bytes
inline size_t nr_scratch_bytes(long long n, long long units_pad, size_t value_size) {
  return (((size_t)n + 64) * value_size + 255) / 256 * 256 + ((size_t)units_pad + 64) * value_size;
}

// The whole fast path.  get(old_id) -> V; reduced: n entries; frontier: n ids (checked to be 0 .. n - 1 by the first kernel).
// Everything is enqueued on the context's stream; the kernels behind the first return at once if *dev_flag == epoch.
// nf == n: the frontier must be 0 .. n - 1 (results by original id).  nf < n (round 6): a strictly ascending subset -- results by frontier
// position; L_in.new_of_old, `offsets` (the ORIGINAL CSR's, for the degree sum -> host_flag[1]) and `pos` are needed.  pos: n words of 64
// bits that ONLY these calls write (zeroed when allocated: an entry is 0 or (epoch of an earlier call, position) -- memory that held
// anything else could show this call's epoch by accident; the epochs of a context never repeat).
// Returns the sequence number to wait for (standard_context_t::mailbox_wait), 0: wait for the stream.
template <typename V, typename Op, typename GetValue>
inline long long nr_full_frontier(const nr_layout_t& L_in, GetValue get, V* reduced, V identity, Op op, standard_context_t& ctx, const int* frontier,
                             long long* host_flag, u32* dev_flag, u32 epoch, long long nf = -1, const int* offsets = nullptr, u64* pos = nullptr) {
  static_assert(sizeof(V) == 4, "the full-frontier neighbour-reduce is instantiated for 4-byte values only (neighborhood.hxx gates on it)");
  hipStream_t s = ctx.stream();
  ++ctx.scratch_epoch;
  nr_layout_t L = L_in;
  V* const vals = (V*)ctx.scratch;
  V* const partial = (V*)((char*)ctx.scratch + (((size_t)L.n + 64) * sizeof(V) + 255) / 256 * 256);
  static unsigned char seen[64] = {};
  if (device_once_t once{seen})
    MGX_HIP(hipFuncSetAttribute((const void*)(k_nr_edges<V, Op, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const bool subset = nf >= 0 && nf < (long long)L.n && pos != nullptr;
  if (subset) {
    L.pos = pos; L.pos_epoch = epoch;
    hipLaunchKernelGGL((k_nr_values_subset<V, GetValue>), dim3(grid_for(L.n, BLOCK, ctx.num_cus * 8)), dim3(BLOCK), 0, s, get, L.old_of_new, L.new_of_old,
                       vals, reduced, identity, (long long)L.n, frontier, nf, offsets, pos, ctx.nr_edges(), host_flag, dev_flag, epoch);
  } else {
    hipLaunchKernelGGL((k_nr_values<V, GetValue>), dim3(grid_for(L.n, BLOCK, ctx.num_cus * 8)), dim3(BLOCK), 0, s, get, L.old_of_new, vals,
                       reduced, identity, (long long)L.n, frontier, host_flag, dev_flag, epoch);
  }
  // (a SUBSET call ends with a one-thread publish kernel -- the degree sum has to reach the host -- and the host spins on its sequence
  //  number; a full-frontier call has nothing to deliver and waits for the stream, as before round 6: with a publish launch behind it
  //  too the call measured 0.277-0.280 ms against 0.266-0.269)
  const long long seq = subset ? ++ctx.mailbox_seq : 0;
  struct publish_t {            // (behind everything else, whichever way the function is left)
    bool subset; hipStream_t s; const u64* e; long long* h; long long* mb; long long seq;
    ~publish_t() { if (subset) hipLaunchKernelGGL(k_nr_publish, dim3(1), dim3(1), 0, s, e, h, mb, seq); }
  } publish{subset, s, ctx.nr_edges(), host_flag + 1, ctx.mailbox, seq};
  if (L.nrs_mu) {
    // the long rows by slice of their destinations + the short rows, one launch; then the fold
    static unsigned char seen_s[64] = {};
    // (U = 8 mini-units in flight per lane measured equal to 4: 116.0 against 116.1 us for the long rows' part)
    if (device_once_t once{seen_s})
      MGX_HIP(hipFuncSetAttribute((const void*)(k_nrs_edges<V, Op, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((k_nrs_edges<V, Op, 1024>), dim3(ctx.num_cus), dim3(1024), nr_lds_bytes(), s, L, (const V*)vals, partial, reduced,
                       identity, op, dev_flag, epoch);
    const u32 grid = L.nrs_tier[0] + (L.nrs_tier[1] - L.nrs_tier[0] + BLOCK / WAVE - 1) / (BLOCK / WAVE) + (L.nrs_tier[2] - L.nrs_tier[1] + BLOCK / 8 - 1) / (BLOCK / 8) +
                     (L.nrs_rows - L.nrs_tier[2] + BLOCK - 1) / BLOCK;
    if (grid && L.nrs_slices + 1u <= (u32)NRS_FOLD_CHUNK)
      hipLaunchKernelGGL((k_nrs_fold<V, Op, false>), dim3(grid), dim3(BLOCK), 0, s, L, (const V*)partial, reduced, identity, op, dev_flag, epoch);
    else if (grid)
      hipLaunchKernelGGL((k_nrs_fold<V, Op, true>), dim3(grid), dim3(BLOCK), 0, s, L, (const V*)partial, reduced, identity, op, dev_flag, epoch);
    return seq;
  }
  const u32 long_rows = L.vs_v[0];
  const bool has_long = L.ub_units > 0 && long_rows > 0, has_short = L.vs_v[3] > L.vs_v[0];
  if (has_long || has_short)
    hipLaunchKernelGGL((k_nr_edges<V, Op, 1024>), dim3(ctx.num_cus), dim3(1024), nr_lds_bytes(), s, L, (const V*)vals, partial, reduced,
                       identity, op, dev_flag, epoch);
  if (has_long) {
    const u32 rest = long_rows > L.big_rows ? long_rows - L.big_rows : 0u;
    const u32 grid = L.big_rows + (rest + BLOCK - 1) / BLOCK;
    if (grid) hipLaunchKernelGGL((k_nr_fold<V, Op>), dim3(grid), dim3(BLOCK), 0, s, L, (const V*)partial, reduced, identity, op, long_rows, dev_flag, epoch);
  }
  return seq;
}

}  // namespace mgx
