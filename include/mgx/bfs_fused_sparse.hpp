// mgx/bfs_fused_sparse.hpp -- a SPARSE level on a rank of the partitioned traversal (bfs_dist2.hpp): no marks to sweep.
//
// A rank turns a level's marks into its discoveries with a sweep over the mark bytes of ALL vertices (k_d2_newbits: 64 MB on
// RMAT-26, ~24 us whatever the level found), and a traversal has more small levels than big ones: the source's, the one
// behind it, the stragglers at the end -- and on every rank but the owner's, a level 0 with nothing at all.  When the rank's
// frontier of the level holds no more edges than its id list has room for (bfs_d2_level_appends: grid-uniform, from the ring
// entry of the level), its discoveries cannot overflow the list, so the push launch writes them THERE, directly:
//   * one wave per 64 entries of a long row, sixteen lanes per short row -- the level is a few thousand edges;
//   * a neighbour whose bit is not in the visited bitmap and whose mark byte is still 0 gets the byte (later levels' sweeps
//     must see it, as after any level) and is appended: ballot, one add to the list's counter per wave and step.  Two waves
//     may see the byte at 0 together: the id is then listed twice, which the list merge decides once (atomicOr on the bitmap);
//   * the same bit goes into the rank's new-bit map with an atomicOr -- the map is what the level ships if some OTHER rank's
//     list overflowed.  The map is all zero between levels (k_d2_lists_apply clears the words of the rank's own list, the
//     bitmap merge clears the map it consumed), so these few bits are all it holds.
// k_d2_newbits then finds ctrl->d2_append_level == level and returns at once.
#pragma once
#include "bfs_fused.hpp"

namespace mgx {

constexpr int D2_LIST_HEAD = 4;          // header words of an id list: [0] count (may exceed the capacity: overflow), [1..3] unused

// the level's push appends to the rank's id list (stable while the level runs: the ring entry was completed by the merge before)
__device__ __forceinline__ bool bfs_d2_level_appends(const bfs_fused_args_t& a, int level) {
  if (!a.d2_list || !a.d2_newbits) return false;
  const bfs_ctrl_t* const c = a.ctrl;
  const u64 E = (c->cursor[level % 3] & BFS_EMASK) + c->ledges[level % 3];
  return E <= (u64)a.d2_list_cap;
}

template <int NT>
__device__ __forceinline__ void bfs_d2_sparse_body(const bfs_fused_args_t& a, int level, u32 block, u32 nblocks) {
  constexpr int NW = NT / WAVE;
  const bfs_ctrl_t* const c = a.ctrl;
  const u64 cur = c->cursor[level % 3], lcur = c->lcursor[level % 3];
  const u32 nf_s = (u32)(cur >> BFS_VSHIFT), E_s = (u32)(cur & BFS_EMASK);
  const u32 nf_l = (u32)(lcur >> BFS_VSHIFT), E_l = (u32)(lcur & BFS_EMASK);
  if (nf_s == 0u && nf_l == 0u) return;
  const int lane = lane_id();
  const u32 gw = block * NW + threadIdx.x / WAVE, GW = nblocks * NW;
  const int* __restrict__ col = a.col_indices;
  const u32* __restrict__ vis = a.visited;
  unsigned char* const mark = a.mark;
  u32* const list = a.d2_list;
  u32* const bits = a.d2_newbits;
  const u32 cap = a.d2_list_cap;

  // all lanes of the wave call this together (the ballot): lanes with `cand` append their neighbour
  auto append = [&](bool cand, u32 d) {
    const u64 m = __ballot(cand);
    if (m == 0ull) return;
    const int leader = __ffsll((long long)m) - 1;
    u32 base = 0;
    if (lane == leader) base = atomicAdd(&list[0], (u32)__popcll(m));
    base = (u32)__shfl((int)base, leader, WAVE);
    if (cand) {
      const u32 at = base + (u32)__popcll(m & ((1ull << lane) - 1ull));
      if (at < cap) list[D2_LIST_HEAD + at] = d;               // (always: the level has no more edges than the list has room)
      atomicOr(&bits[d >> 5], 1u << (d & 31u));
      mark[d] = 1;
    }
  };
  // Four entries per lane at a time, and their three dependent look-ups -- neighbour id, bitmap word, mark byte -- each issued
  // for all four before the first is looked at: a row costs three round trips to memory, not three per entry (a lane walking
  // its row entry by entry took ~100 us for a level of 20 000 edges).
  auto four = [&](u32 first, u32 step, u32 deg, u32 start, bool ok) {
    u32 d[4], w[4];
    unsigned char mk[4];
    bool v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32 e = first + (u32)j * step;
      v[j] = ok && e < deg;
      d[j] = v[j] ? (u32)col[start + e] : 0u;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = v[j] ? vis[d[j] >> 5] : 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 4; ++j) mk[j] = v[j] ? mark[d[j]] : (unsigned char)1;
#pragma unroll
    for (int j = 0; j < 4; ++j) append(v[j] && !((w[j] >> (d[j] & 31u)) & 1u) && mk[j] == 0, d[j]);
  };

  // long rows: one wave per 64-entry UNIT of the padded rank space the long-row queue is scanned in (bfs_lq_*: every row starts
  // on a multiple of 64, so a unit lies inside one row) -- the frontier behind the source is a handful of hubs, and a wave per
  // ROW walked 30 000 entries in 120 dependent steps (123 us for that level on RMAT-26 / 8).  The row of a unit: bisection
  // over the scanned offsets, every lane the same loads.
  const u32* __restrict__ lrow = a.lq_row[level & 1];
  const u32* __restrict__ loff = a.lq_off[level & 1];
  for (u32 r = gw * WAVE; r < E_l; r += GW * WAVE) {           // (wave-uniform)
    u32 lo = 0, hi = nf_l;
    while (hi - lo > 1u) {
      const u32 mid = (lo + hi) >> 1;
      if ((loff[mid] & ~63u) <= r) lo = mid; else hi = mid;
    }
    const u32 e0 = loff[lo];
    const u32 deg = bfs_lq_degree(e0, lo + 1u < nf_l ? loff[lo + 1u] : E_l);
    const u32 e = r - (e0 & ~63u) + (u32)lane;                 // this lane's entry of the row
    const bool v = e < deg;
    const u32 d = v ? (u32)col[lrow[lo] + e] : 0u;
    const u32 w = v ? vis[d >> 5] : 0xFFFFFFFFu;
    const unsigned char mk = v ? mark[d] : (unsigned char)1;
    append(v && !((w >> (d & 31u)) & 1u) && mk == 0, d);
  }
  // short rows: sixteen lanes per row, four rows per wave step, 64 entries of each per step
  const u32* __restrict__ srow = a.fr_row[level & 1];
  const u32* __restrict__ soff = a.fr_off[level & 1];
  const u32 sub = (u32)lane & 15u, grp = (u32)lane >> 4;
  for (u32 i0 = gw * 4u; i0 < nf_s; i0 += GW * 4u) {           // (wave-uniform)
    const u32 i = i0 + grp;
    const bool ok = i < nf_s;
    const u32 start = ok ? srow[i] : 0u;
    const u32 o0 = ok ? soff[i] : 0u;
    const u32 o1 = ok ? (i + 1u < nf_s ? soff[i + 1u] : E_s) : 0u;
    const u32 deg = o1 - o0;
    u32 longest = deg;
#pragma unroll
    for (int sh = 16; sh < WAVE; sh <<= 1) {
      const u32 o = (u32)__shfl_xor((int)longest, sh, WAVE);
      longest = o > longest ? o : longest;
    }
    for (u32 e0 = 0; e0 < longest; e0 += 64u) four(e0 + sub, 16u, deg, start, ok);
  }
}

}  // namespace mgx
