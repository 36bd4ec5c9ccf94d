// mgx/bfs_fused_vshort.hpp -- push over the SHORT rows of a level, vertex by vertex.
//
// bfs_fused_wave.hpp addresses the short-row queue by edge RANK: stage 64 queue entries in LDS, resolve every rank to
// its row (a multiply-high when the staged rows share a degree, a binary search otherwise), gather.  RMAT-22 keeps 13 %
// of its edges in rows of fewer than 64 -- and spends 32 % of its push time there (110 of 346 us per traversal).  The
// levels that matter have MOST short rows in their frontier (the big level: 3 of 4 short-row edges; the one behind it:
// two thirds of the vertices of degree < 16), and the hub-first layout is sorted by degree, so 64 consecutive vertices
// have (nearly) one degree.  Then nothing needs a queue or a search:
//
//   * a wave step takes 64 / LPR consecutive vertices, LPR lanes per vertex: 16 lanes for degrees 17..63, 4 for 5..16,
//     1 for 1..4 -- the class follows from the vertex id (three boundaries found when the layout is built);
//   * per lane: the row's two offsets (one 8-byte load, neighbouring lanes neighbouring addresses), the vertex's bit in
//     the frontier bitmap k_bfs_build left behind, then ONE 16-byte load of the lane's four entries (entries 4 sub ..
//     4 sub + 3 of the row): the rows of consecutive vertices are contiguous in the CSR, so a wave instruction reads one
//     contiguous stretch.  Lanes of vertices outside the frontier (and lanes past a row's end) read four -1 instead;
//   * the visited test is the one of bfs_fused_dense.hpp (sentinel words around the LDS bitmap, a miss claims its bit);
//   * steps are interleaved over all waves of the grid; the loads of step t + 1 are in flight while step t is tested,
//     the offsets and frontier bits of step t + 2 while those are issued.
// Chosen per slot like the unit blocks (bfs_short_is_dense): the frontier bitmap must be current and the level must hold
// at least 1 / vshort_div of all short-row edges; otherwise the queue search of bfs_fused_wave.hpp runs.
#pragma once
#include "bfs_fused.hpp"
#include "bfs_fused_dense.hpp"

namespace mgx {

constexpr size_t bfs_vshort_lds_bytes(int hotw) { return (size_t)hotw * 4 + 128; }

__device__ __forceinline__ bool bfs_short_is_dense(const bfs_fused_args_t& a, const bfs_ctrl_t* c, int slot, u64 cur) {
  if (a.vs_div == 0u || c->fb_slot != slot) return false;
  return (cur & BFS_EMASK) * (u64)a.vs_div >= (u64)a.vs_edges;
}

struct __attribute__((aligned(4))) bfs_u32x4u { u32 x, y, z, w; };   // 16-byte load at 4-byte alignment

// the vertex-by-vertex pass of one workgroup (block `block` of `nblocks`) over an LDS prefix that is already set up
// (bfs_hot_setup in bfs_fused_dense.hpp)
template <int NT, int HOTW>
__device__ __forceinline__ void bfs_vshort_work(const bfs_fused_args_t& a, u32* const hot, u32 hot_n, u32 defer_n, u32 block,
                                                u32 nblocks, int& marks) {
  constexpr int NW = NT / WAVE;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
  const int lane = lane_id();
  unsigned char* __restrict__ mark = a.mark;
  const int* __restrict__ col = a.vs_col ? a.vs_col : a.col_indices;
  const u32* __restrict__ ro = a.row_offsets;
  const u32* __restrict__ fbits = a.d2_front ? a.d2_front : a.frontier_bits;
  // classes: [vs_v[0], vs_v[1]) 16 lanes per vertex, [vs_v[1], vs_v[2]) 4, [vs_v[2], vs_v[3]) 1
  const u32 b0 = a.vs_v[0], b1 = a.vs_v[1], b2 = a.vs_v[2], b3 = a.vs_v[3];
  // the widest class: 16 lanes per vertex cover 64 entries; with a long-row threshold of 32 or less its rows have at most 31
  // entries and 8 lanes (32 entries) do -- twice the vertices per wave step
  const u32 shift0 = (a.long_min > 0 && a.long_min <= 32) ? 3u : 4u;
  const u32 vps0 = 64u >> shift0;
  // ... and the class of degrees 5 .. 16 is split where the degrees drop below 9 (a.vs_v9): four lanes per vertex above, two below
  const u32 b9 = a.vs_v9 >= b1 && a.vs_v9 <= b2 ? a.vs_v9 : b2;
  const u32 s16 = (b1 - b0 + vps0 - 1u) / vps0, s4 = (b9 - b1 + 15u) / 16u, s2 = (b2 - b9 + 31u) / 32u, s1 = (b3 - b2 + 63u) / 64u;
  const u32 T = s16 + s4 + s2 + s1;                             // wave steps in all
  const u32 W = nblocks * NW, w = block * NW + (u32)wave;
  const u32 dummy = a.vs_dummy;                                 // index into col of four readable entries behind the CSR

  constexpr int K = 2;                                          // wave steps per pipeline stage
  struct raw_t { u32 lo, hi, fw, v_sub; };                      // what a step's planning loads return: row ends, frontier WORD (raw), bit | sub-position
  struct step_t { u32 e0, cnt; };                               // first entry of the lane's four (dummy if none), how many are real
  // stage A: the planning loads of step s.  NOTHING is looked at yet and no load sits under a condition: a value that is
  // shifted or masked here would have to land here (the compiler waits where the ALU op stands), and a conditional load
  // makes it drain everything in flight (bfs_fused.hpp, "countable loads") -- either way the entry loads issued a moment
  // ago would be waited for before the previous step is tested.  Lanes without a vertex read vertex 0 and carry sub = ~0.
  auto plan_load = [&](u32 s) -> raw_t {
    raw_t r;
    u32 lpr_shift, vbase, vend;
    if (s < s16) { lpr_shift = shift0; vbase = b0 + s * vps0; vend = b1; }
    else if (s < s16 + s4) { lpr_shift = 2; vbase = b1 + (s - s16) * 16u; vend = b9; }
    else if (s < s16 + s4 + s2) { lpr_shift = 1; vbase = b9 + (s - s16 - s4) * 32u; vend = b2; }
    else { lpr_shift = 0; vbase = b2 + (s - s16 - s4 - s2) * 64u; vend = b3; }
    const u32 v = vbase + ((u32)lane >> lpr_shift);
    const u32 sub = (u32)lane & ((1u << lpr_shift) - 1u);
    const bool in = s < T && v < vend;
    const u32 vc = in ? v : 0u;
    const bfs_u32x2 ext = *(const bfs_u32x2*)(ro + vc);
    r.lo = ext.x; r.hi = ext.y;
    r.fw = fbits[vc >> 5];
    r.v_sub = in ? (sub | ((vc & 31u) << 8)) : 0xFFFFFFFFu;
    return r;
  };
  // stage B: from the landed planning loads to the lane's entry range
  auto resolve = [&](const raw_t& r) -> step_t {
    step_t p; p.e0 = dummy; p.cnt = 0;
    const u32 deg = r.hi - r.lo;
    const u32 sub = r.v_sub & 0xFFu;
    const bool in_frontier = r.v_sub != 0xFFFFFFFFu && ((r.fw >> ((r.v_sub >> 8) & 31u)) & 1u);
    if (in_frontier && sub * 4u < deg) {
      p.e0 = r.lo + sub * 4u;
      const u32 left = deg - sub * 4u;
      p.cnt = left < 4u ? left : 4u;
    }
    return p;
  };
  auto probe = [&](u32 d) -> u32 {
    int idx = (int)d >> 5;
    idx = idx < -1 ? -1 : idx;
    idx = idx > HOTW ? HOTW : idx;
    return hot[idx];
  };
  auto decide = [&](u32 d, u32 wd) {
    if (!((wd >> (d & 31u)) & 1u)) {
      const u32 bit = 1u << (d & 31u);
      bool is_new = true;
      if (d < hot_n) is_new = !(atomicOr(&hot[d >> 5], bit) & bit);
      if (is_new) { if (d >= defer_n) mark[d] = 1; ++marks; }
    }
  };

  if (w < T) {
    // iteration i handles steps w + (K i + k) W: planning loads of iteration i + 2 and entry loads of iteration i + 1
    // are in flight while iteration i is tested
    raw_t rA[K];
    bfs_u32x4u dL[K], dT[K];
    u32 cL[K], cT[K];
#pragma unroll
    for (int k = 0; k < K; ++k) rA[k] = plan_load(w + (u32)k * W);
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const step_t p = resolve(rA[k]);
      dL[k] = *(const bfs_u32x4u*)(col + p.e0);
      cL[k] = p.cnt;
    }
#pragma unroll
    for (int k = 0; k < K; ++k) rA[k] = plan_load(w + (u32)(K + k) * W);
    for (u32 s = w; s < T; s += K * W) {
#pragma unroll
      for (int k = 0; k < K; ++k) { dT[k] = dL[k]; cT[k] = cL[k]; }
      step_t p[K];
#pragma unroll
      for (int k = 0; k < K; ++k) p[k] = resolve(rA[k]);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        dL[k] = *(const bfs_u32x4u*)(col + p[k].e0);            // (past the last step: the dummy)
        cL[k] = p[k].cnt;
      }
#pragma unroll
      for (int k = 0; k < K; ++k) rA[k] = plan_load(s + (u32)(2 * K + k) * W);
#pragma unroll
      for (int k = 0; k < K; ++k) {
        // entries past the lane's count (the next row's) read as visited
        const u32 d0 = cT[k] > 0u ? dT[k].x : 0xFFFFFFFFu, d1 = cT[k] > 1u ? dT[k].y : 0xFFFFFFFFu;
        const u32 d2 = cT[k] > 2u ? dT[k].z : 0xFFFFFFFFu, d3 = cT[k] > 3u ? dT[k].w : 0xFFFFFFFFu;
        const u32 w0 = probe(d0), w1 = probe(d1), w2 = probe(d2), w3 = probe(d3);
        decide(d0, w0); decide(d1, w1); decide(d2, w2); decide(d3, w3);
      }
    }
  }
}

// The same with the COLD TEST of a big graph (most endpoints behind the LDS prefix: a rank of a partitioned RMAT-26): a neighbour
// behind the prefix is looked up in the global bitmap before it is marked -- the words of a step's cold entries are requested
// when the entries land and looked at one step later, so three steps are in flight: entries (s + 1), bitmap words (s), test
// (s - 1).  One wave step per iteration (the words cost the registers the second step had).
template <int NT, int HOTW>
__device__ __forceinline__ void bfs_vshort_work_coldtest(const bfs_fused_args_t& a, u32* const hot, u32 hot_n, u32 defer_n, u32 block,
                                                         u32 nblocks, int& marks) {
  constexpr int NW = NT / WAVE;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
  const int lane = lane_id();
  unsigned char* __restrict__ mark = a.mark;
  const int* __restrict__ col = a.vs_col ? a.vs_col : a.col_indices;
  const u32* __restrict__ ro = a.row_offsets;
  const u32* __restrict__ fbits = a.d2_front ? a.d2_front : a.frontier_bits;
  const u32* __restrict__ vis = a.visited;
  const u32 b0 = a.vs_v[0], b1 = a.vs_v[1], b2 = a.vs_v[2], b3 = a.vs_v[3];
  const u32 shift0 = (a.long_min > 0 && a.long_min <= 32) ? 3u : 4u;
  const u32 vps0 = 64u >> shift0;
  const u32 b9 = a.vs_v9 >= b1 && a.vs_v9 <= b2 ? a.vs_v9 : b2;
  const u32 s16 = (b1 - b0 + vps0 - 1u) / vps0, s4 = (b9 - b1 + 15u) / 16u, s2 = (b2 - b9 + 31u) / 32u, s1 = (b3 - b2 + 63u) / 64u;
  const u32 T = s16 + s4 + s2 + s1;
  const u32 W = nblocks * NW, w = block * NW + (u32)wave;
  const u32 dummy = a.vs_dummy;
  struct raw_t { u32 lo, hi, fw, v_sub; };
  struct step_t { u32 e0, cnt; };
  auto plan_load = [&](u32 s) -> raw_t {
    raw_t r;
    u32 lpr_shift, vbase, vend;
    if (s < s16) { lpr_shift = shift0; vbase = b0 + s * vps0; vend = b1; }
    else if (s < s16 + s4) { lpr_shift = 2; vbase = b1 + (s - s16) * 16u; vend = b9; }
    else if (s < s16 + s4 + s2) { lpr_shift = 1; vbase = b9 + (s - s16 - s4) * 32u; vend = b2; }
    else { lpr_shift = 0; vbase = b2 + (s - s16 - s4 - s2) * 64u; vend = b3; }
    const u32 v = vbase + ((u32)lane >> lpr_shift);
    const u32 sub = (u32)lane & ((1u << lpr_shift) - 1u);
    const bool in = s < T && v < vend;
    const u32 vc = in ? v : 0u;
    const bfs_u32x2 ext = *(const bfs_u32x2*)(ro + vc);
    r.lo = ext.x; r.hi = ext.y;
    r.fw = fbits[vc >> 5];
    r.v_sub = in ? (sub | ((vc & 31u) << 8)) : 0xFFFFFFFFu;
    return r;
  };
  auto resolve = [&](const raw_t& r) -> step_t {
    step_t p; p.e0 = dummy; p.cnt = 0;
    const u32 deg = r.hi - r.lo;
    const u32 sub = r.v_sub & 0xFFu;
    const bool in_frontier = r.v_sub != 0xFFFFFFFFu && ((r.fw >> ((r.v_sub >> 8) & 31u)) & 1u);
    if (in_frontier && sub * 4u < deg) {
      p.e0 = r.lo + sub * 4u;
      const u32 left = deg - sub * 4u;
      p.cnt = left < 4u ? left : 4u;
    }
    return p;
  };
  // (unconditional: entries inside the prefix and lanes without an entry ask for word 0)
  auto cold_word = [&](u32 d) -> u32 { return vis[(d >= hot_n && d != 0xFFFFFFFFu) ? (d >> 5) : 0u]; };
  auto decide = [&](u32 d, u32 wcold) {
    if (d == 0xFFFFFFFFu) return;
    const u32 bit = 1u << (d & 31u);
    bool is_new;
    if (d < hot_n) is_new = !(hot[d >> 5] & bit) && !(atomicOr(&hot[d >> 5], bit) & bit);
    else is_new = !(wcold & bit);
    if (is_new) { if (d >= defer_n) mark[d] = 1; ++marks; }
  };
  if (w < T) {
    raw_t rA = plan_load(w);
    step_t p = resolve(rA);
    bfs_u32x4u dL = *(const bfs_u32x4u*)(col + p.e0);
    u32 cL = p.cnt;
    rA = plan_load(w + W);
    u32 dP[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, wP[4] = {0u, 0u, 0u, 0u};
    for (u32 s = w; s < T; s += W) {
      const bfs_u32x4u dT = dL;
      const u32 cT = cL;
      p = resolve(rA);
      dL = *(const bfs_u32x4u*)(col + p.e0);                    // (past the last step: the dummy)
      cL = p.cnt;
      rA = plan_load(s + 2u * W);
      // entries past the lane's count (the next row's) do not exist
      const u32 dN[4] = {cT > 0u ? dT.x : 0xFFFFFFFFu, cT > 1u ? dT.y : 0xFFFFFFFFu, cT > 2u ? dT.z : 0xFFFFFFFFu, cT > 3u ? dT.w : 0xFFFFFFFFu};
      u32 wN[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) wN[j] = cold_word(dN[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) decide(dP[j], wP[j]);
#pragma unroll
      for (int j = 0; j < 4; ++j) { dP[j] = dN[j]; wP[j] = wN[j]; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) decide(dP[j], wP[j]);
  }
}

template <int NT, int HOTW, bool COLDT = false>
__device__ __forceinline__ void bfs_vshort_body(const bfs_fused_args_t& a, int slot, u32 block, u32 nblocks, int stat_level,
                                                bool cold = false) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_int;
  u32* const hot = bfs_hot_setup<NT, HOTW>(a, smem, &s_int, cold ? 0xFFFFFFFFu : 0u);    // cold: the cold-edge pass has those entries
  const int lane = lane_id();
  bfs_ctrl_t* const c = a.ctrl;
  const u32 hot_n = ((u32)a.n < (u32)(HOTW * 32)) ? (u32)a.n : (u32)(HOTW * 32);
  const u32 defer_n = bfs_defer_limit(a, hot_n);
  int marks = 0;
  if constexpr (COLDT) bfs_vshort_work_coldtest<NT, HOTW>(a, hot, hot_n, defer_n, block, nblocks, marks);
  else bfs_vshort_work<NT, HOTW>(a, hot, hot_n, defer_n, block, nblocks, marks);
  (void)bfs_hot_epilogue<NT>(a, hot, (defer_n + 31u) >> 5, slot, s_int + 4, marks);
  bfs_body_finish(a, marks, slot, stat_level, s_int);
}


}  // namespace mgx
