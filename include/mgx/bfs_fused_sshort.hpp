// mgx/bfs_fused_sshort.hpp -- push over the SHORT rows of a level as ONE stream of their entries.
//
// The vertex-by-vertex body (bfs_fused_vshort.hpp) walks the short rows by vertex: per vertex a load of its two offsets,
// its frontier bit, then 16 bytes of entries on 16, 4 or 1 lanes -- three dependent trips per step, a wave gets six
// steps, and a vertex of degree 1 uses a quarter of what its lane loads: the 18 M short-row edges of RMAT-22's big level
// took 36 us next to 72 us for five times as many long-row edges.
//
// Under the degree-sorted layout the rows of ONE degree are consecutive, and so are their entries: the short rows' part
// of col_indices is 63 regions (degrees 63 .. 1), and inside the region of degree d entry e belongs to row
//     first_row(d) + (e - first_entry(d)) / d
// -- no offsets, no owner table, no search.  So the short rows are read like the unit blocks: every wave takes a
// contiguous run of 256-entry chunks (16 aligned bytes per lane, every entry loaded exactly once, all lanes busy
// whatever the degree); a lane computes the rows of its four entries (a multiply-high by the degree's reciprocal, the
// region is wave-uniform except where a chunk crosses into the next one), asks the frontier bitmap for them
// (neighbouring lanes, neighbouring or equal words) and tests the entries of frontier rows against the LDS prefix as
// everywhere else.  Nothing here depends on loaded data except the test itself: the entries of chunk t + 2 and the
// frontier words of chunk t + 1 are in flight while chunk t is tested.  Rows outside the frontier cost their bytes
// (a quarter of them on the big level) and nothing else.
// Chosen like the vertex-by-vertex body (bfs_short_is_dense), which it replaces when MGX_BFS_SSTREAM=1 asks for it.
// MEASURED (RMAT-22): 40 / 37 us for the short rows of the big level and of the level behind it, against 36 / 31 us
// vertex by vertex -- every entry pays a row computation (multiply-high, two corrections) and a frontier-word gather of
// its own where the other body pays one per VERTEX, and the bytes it saves were never what bounded those 18 M edges.
// Off by default; kept because the lookup it needed exposed the serialised prefix copy (bfs_copy_prefix, bfs_fused.hpp).
#pragma once
#include "bfs_fused.hpp"
#include "bfs_fused_dense.hpp"

namespace mgx {

constexpr int BFS_SS_MAXDEG = 64;                  // table slots: degrees 0 .. 64
// ss_tab layout: [0 .. 64] first entry of the region of degree d (TE[0] = the end of the short rows' entries),
//                [65 .. 129] first row of the region of degree d
constexpr int BFS_SS_TAB_WORDS = 2 * (BFS_SS_MAXDEG + 1);

template <int NT, int HOTW>
__device__ __forceinline__ void bfs_sstream_work(const bfs_fused_args_t& a, u32* const hot, const u32* const tab, u32 hot_n, u32 defer_n,
                                                 u32 block, u32 nblocks, int& marks) {
  constexpr int NW = NT / WAVE;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
  const int lane = lane_id();
  unsigned char* __restrict__ mark = a.mark;
  const int* __restrict__ col = a.col_indices;
  const u32* __restrict__ fbits = a.frontier_bits;
  // (the table sits in LDS, behind the prefix: a lane that has to look up another region must not do it with a global load
  //  under a condition -- the compiler would drain every load in flight at the next use of one, bfs_fused.hpp "countable loads")
  const u32* const TE = tab;                                     // first entry of region d
  const u32* const TR = tab + (BFS_SS_MAXDEG + 1);               // first row of region d
  const int D = a.ss_dmax;                                       // the largest short degree (long_min - 1)
  const u32 e_begin = TE[D], e_end = TE[0];
  if (e_end <= e_begin) return;
  const u32 c_begin = e_begin >> 8, c_end = (e_end + 255u) >> 8;          // 256-entry chunks, aligned from entry 0
  const u32 W = nblocks * NW, w = block * NW + (u32)wave;
  const u32 per = (c_end - c_begin + W - 1u) / W;
  const u32 c0 = c_begin + w * per;
  if (c0 >= c_end) return;
  const u32 c1 = c0 + per < c_end ? c0 + per : c_end;
  const u32 dummy = a.vs_dummy;                                  // four entries of -1 behind the CSR

  // the region of the wave's first entry (wave-uniform walk: the regions follow each other as the degree falls)
  int d_w = D;
  {
    const u32 first = (c0 << 8) > e_begin ? (c0 << 8) : e_begin;
    while (d_w > 1 && TE[d_w - 1] <= first) --d_w;
  }
  u32 te_w = TE[d_w], tr_w = TR[d_w], hi_w = TE[d_w - 1];
  u32 rc_w = 0xFFFFFFFFu / (u32)d_w;               // (+ 1 in 64 bits: the reciprocal of the region's degree)

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
  // the row of entry e (e_begin <= e < e_end)
  auto row_of = [&](u32 e) -> u32 {
    int d = d_w;
    u32 te = te_w, tr = tr_w, hi = hi_w, rc = rc_w;
    while (e >= hi) {                           // (rare: the chunk crosses into the next region(s); LDS reads only)
      --d;
      te = hi; tr = TR[d]; hi = TE[d - 1];
      rc = 0xFFFFFFFFu / (u32)d;
    }
    const u32 x = e - te;
    u32 q = (u32)(((u64)x * ((u64)rc + 1ull)) >> 32);                      // x / d, one off at most (x < 2^31)
    if ((q + 1u) * (u32)d <= x) ++q;
    if (q * (u32)d > x) --q;
    return tr + q;
  };
  // stage A: the 16 bytes of the lane's four entries of chunk c (dummy: four -1)
  auto load_entries = [&](u32 c) -> bfs_u32x4 {
    const u32 e0 = (c << 8) + (u32)lane * 4u;
    const bool any = c < c1 && e0 + 4u > e_begin && e0 < e_end;
    return *(const bfs_u32x4*)(col + (any ? e0 : dummy));
  };
  // stage B: the frontier words of the lane's four entries of chunk c and the bit positions inside them (nothing is
  // looked at yet: the words are still in flight when this returns; entries outside the short rows ask for word 0, bit 32)
  struct act_t { u32 w[4]; u32 sh[4]; };
  auto activity = [&](u32 c) -> act_t {
    act_t r;
    const u32 e0 = (c << 8) + (u32)lane * 4u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const u32 e = e0 + (u32)j;
      const bool in = c < c1 && e >= e_begin && e < e_end;
      const u32 row = in ? row_of(e) : 0u;
      r.w[j] = fbits[row >> 5];
      r.sh[j] = in ? (row & 31u) : 32u;
    }
    return r;
  };
  auto active = [&](const act_t& r, int j) -> bool { return r.sh[j] < 32u && ((r.w[j] >> r.sh[j]) & 1u); };
  auto advance_region = [&](u32 c) {           // wave-uniform: the region of the first entry of chunk c
    const u32 first = (c << 8) > e_begin ? (c << 8) : e_begin;
    while (d_w > 1 && hi_w <= first) { --d_w; te_w = hi_w; tr_w = TR[d_w]; hi_w = TE[d_w - 1]; rc_w = 0xFFFFFFFFu / (u32)d_w; }
  };

  bfs_u32x4 e1 = load_entries(c0), e2 = load_entries(c0 + 1u);
  act_t a1 = activity(c0);
  for (u32 c = c0; c < c1; ++c) {
    const bfs_u32x4 eT = e1;
    const act_t aT = a1;
    e1 = e2;
    e2 = load_entries(c + 2u);
    advance_region(c + 1u);
    a1 = activity(c + 1u);
    const u32 d0 = active(aT, 0) ? eT.x : 0xFFFFFFFFu, d1 = active(aT, 1) ? eT.y : 0xFFFFFFFFu;
    const u32 d2 = active(aT, 2) ? eT.z : 0xFFFFFFFFu, d3 = active(aT, 3) ? eT.w : 0xFFFFFFFFu;
    const u32 w0 = probe(d0), w1 = probe(d1), w2 = probe(d2), w3 = probe(d3);
    decide(d0, w0); decide(d1, w1); decide(d2, w2); decide(d3, w3);
  }
}

// HOTW: words of the LDS prefix of THIS body -- BFS_SS_TAB_PAD words fewer than the other bodies', for the table behind it
constexpr int BFS_SS_TAB_PAD = 160;
template <int NT, int HOTW>
__device__ __forceinline__ void bfs_sstream_body(const bfs_fused_args_t& a, int slot, u32 block, u32 nblocks, int stat_level,
                                                 bool cold = false) {
  static_assert(BFS_SS_TAB_WORDS <= BFS_SS_TAB_PAD - 16, "the region table fits behind the prefix");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_int;
  u32* const hot = bfs_hot_setup<NT, HOTW>(a, smem, &s_int, cold ? 0xFFFFFFFFu : 0u);
  u32* const tab = (u32*)(s_int + 16);                     // (s_int: 16 ints of bookkeeping behind the prefix)
  for (int i = threadIdx.x; i < BFS_SS_TAB_WORDS; i += NT) tab[i] = a.ss_tab[i];
  __syncthreads();
  const u32 hot_n = ((u32)a.n < (u32)(HOTW * 32)) ? (u32)a.n : (u32)(HOTW * 32);
  const u32 defer_n = bfs_defer_limit(a, hot_n);
  int marks = 0;
  bfs_sstream_work<NT, HOTW>(a, hot, tab, hot_n, defer_n, block, nblocks, marks);
  (void)bfs_hot_epilogue<NT>(a, hot, (defer_n + 31u) >> 5, slot, s_int + 4, marks);
  bfs_body_finish(a, marks, slot, stat_level, s_int);
}

}  // namespace mgx
