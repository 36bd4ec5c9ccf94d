// mgx/bfs_fused_sliced.hpp -- push over the LONG rows of a HUB level, by slice of the destinations (round 6).
//
// A hub level -- the few thousand rows that hold 10-50 % of a skewed graph's edges, two or three levels into a traversal -- finds
// nearly every neighbour unvisited.  The other bodies keep the bitmap's 652 288-vertex prefix per workgroup in LDS and claim there;
// on such a level every one of 512 workgroups ends up with its own copy of (nearly) the same ten thousand claims, which it then
// FLUSHES as an 80 KB bitmap for the queue build to OR: 41-74 MB written and read again, 13-16 us of the level (DESIGN 3.1,
// "Round 6"), on top of 1 024 copies of the prefix into LDS.
//
// Here the level's long rows are read from the layout's long rows BY SLICE OF THEIR DESTINATIONS (mgx_layout.hip:
// mgx_nrs_build_device -- the neighbour-reduce's layout, mgx/nreduce.hpp): slice k holds the entries with destination in
// [k S, (k + 1) S), S = NR_HOTV = 39 936 = 39 runs of 1 024 vertices, row after row, 8 offsets of 16 bits per 16-byte mini-unit;
// nrs_off[k rows + r] says where row r's mini-units of slice k start.  A workgroup belongs to ONE slice:
//   * it copies that slice's 1 248 words of the bitmap into LDS (5 KB instead of 80);
//   * its waves take equal pieces of the level's long-row queue by PADDED edge rank -- the split the queue walk uses -- and a piece
//     of a row that spans several workgroups is cut in proportion (cut() below: the same arithmetic on both sides of a boundary, so
//     the pieces tile the row's mini-units exactly);
//   * a wave stages up to 64 rows at a time -- row start and padded offsets from the queue, the row's id from its start (long rows
//     are at least 32 entries apart: start >> 5 names the row, args.nrs_vid_of), its range of mini-units from nrs_off -- scans the
//     counts and walks the mini-units 64 at a time, a lane a mini-unit: one 16-byte load, eight LDS probes, a claim (ds_or) for
//     every bit that is not set;
//   * at its end the workgroup writes its 1 248 words to ITS place in args.slice_flush -- 5 KB -- and k_bfs_build2 ORs the J_k
//     buffers of a slice into the runs of that slice (the shape of its loop over the flush buffers: 512 x 5 KB = 2.5 MB in all).
// The TAIL behind the last hot slice (6 % of RMAT-22's long-row entries: 4 ids of 32 bits per mini-unit) is marked as the queue
// walk marks cold neighbours: a byte store, untested (the cold-edge pass takes what lies behind the 652 288-vertex prefix when the
// slot runs it).  No marks for the hot slices at all, no flush of 80 KB bitmaps, no prefix copy.
#pragma once
#include "bfs_fused.hpp"
#include "nreduce.hpp"

namespace mgx {

constexpr int BFS_SL_WORDS = NR_HOTV / 32;              // words of a slice's bitmap (1 248)
static_assert(NR_HOTV == 39936, "a slice is 39 whole runs of 1 024 vertices: k_bfs_build2 (bfs_fused.hpp) ORs the slices' buffers run by run with that number");
constexpr int BFS_SL_RUNS = NR_HOTV / 1024;             // 39
constexpr size_t bfs_sliced_lds_bytes() { return (size_t)(BFS_SL_WORDS + 8) * 4 + (size_t)(1024 / WAVE) * 2 * WAVE * 4 + 64; }

// the slot's long rows go through the sliced body (grid-uniform: queue sizes and ctrl->reached are stable while the slot runs)
__device__ __forceinline__ bool bfs_level_is_sliced(const bfs_fused_args_t& a, const bfs_ctrl_t* c, int slot, u64 lcur) {
  if (!a.nrs_mu || !a.slice_flush || (lcur >> BFS_VSHIFT) == 0) return false;
  if (c->lazy_slot == slot) return false;                               // (no queues)
  if (c->ledges[slot % 3] < (u64)a.sliced_min_edges) return false;
  const u32 range = (u32)a.n < a.defer_words * 32u ? (u32)a.n : a.defer_words * 32u;
  return c->reached * (u64)a.defer_reach_mul < (u64)range * (u64)a.defer_reach_div;     // the level defers its hot marks (bfs_defer_limit)
}

template <int NT>
__device__ __forceinline__ void bfs_sliced_body(const bfs_fused_args_t& a, int slot, u32 block, u32 nblocks, int stat_level, bool cold_pass_runs) {
  constexpr int NW = NT / WAVE;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32* const bits = (u32*)smem;                                          // [BFS_SL_WORDS] + the word a padding offset reads
  int* const s_int = (int*)(bits + BFS_SL_WORDS + 4);
  u32* const s_pre = (u32*)(s_int + 4);                                  // [NW][64] exclusive counts
  u32* const s_m0 = s_pre + NW * WAVE;                                   // [NW][64] first mini-unit
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
  const int lane = lane_id();
  const bfs_ctrl_t* const c = a.ctrl;
  const u64 cur = c->lcursor[slot % 3];
  const u32 nf = (u32)(cur >> BFS_VSHIFT);
  const u32 E = (u32)(cur & BFS_EMASK);                                  // padded edges: a multiple of 64
  // my slice, my part of it
  int k = 0;
  const int K1 = (int)a.nrs_slices + 1;                                  // hot slices + the tail
  while (k + 1 < K1 && block >= (u32)a.sl_base[k + 1]) ++k;
  const u32 part = block - (u32)a.sl_base[k], parts = (u32)a.sl_base[k + 1] - (u32)a.sl_base[k];
  const bool tail = k == (int)a.nrs_slices;
  if (block >= (u32)a.sl_base[K1] || parts == 0u) return;                // (more workgroups than the table deals)
  const u32 v0 = (u32)k * (u32)NR_HOTV;
  if (!tail) {
    const u32 nwords = ((u32)a.n + 31u) >> 5;
    for (u32 i = threadIdx.x; i < (u32)BFS_SL_WORDS; i += NT) {
      const u32 w = v0 / 32u + i;
      bits[i] = w < nwords ? a.visited[w] : 0xFFFFFFFFu;
    }
    if (threadIdx.x < 4) bits[BFS_SL_WORDS + threadIdx.x] = 0xFFFFFFFFu;   // a padding offset (NR_HOTV) reads "visited"
  }
  if (threadIdx.x == 0) s_int[0] = 0;
  __syncthreads();
  int marks = 0;
  // this wave's piece of the queue, by padded edge rank
  const u64 W = (u64)parts * NW, w = (u64)part * NW + (u64)wave;
  const u32 lo = (u32)(((u64)E * w / W) & ~63ull), hi = w + 1 == W ? E : (u32)(((u64)E * (w + 1) / W) & ~63ull);
  if (lo < hi && nf) {
    const u32* __restrict__ q_row = a.lq_row[slot & 1];
    const u32* __restrict__ q_off = a.lq_off[slot & 1];
    const u32* __restrict__ off = a.nrs_off + (size_t)k * a.nrs_rows;
    const uint4* __restrict__ mu = (const uint4*)a.nrs_mu;
    unsigned char* __restrict__ mark = a.mark;
    u32* const pre = s_pre + wave * WAVE;
    u32* const m0s = s_m0 + wave * WAVE;
    u32 seg = (u32)(wave_upper_bound(q_off, (long long)nf, lo + 63u) - 1);      // the row that holds padded rank lo
    for (;;) {
      // ---- stage up to 64 rows: lane l takes row seg + l
      const u32 i = seg + (u32)lane;
      const bool in_q = i < nf;
      const u32 e0 = q_off[in_q ? i : nf - 1u];
      const u32 e1 = (i + 1u < nf) ? q_off[i + 1u] : E;
      const u32 start = q_row[in_q ? i : nf - 1u];
      const u32 P = e0 & ~63u, P1 = (in_q && i + 1u < nf) ? (e1 & ~63u) : E;
      const bool live = in_q && P < hi;
      const u32 vid = a.nrs_vid_of[live ? (start >> 5) : 0u];
      const u32 ma = off[live ? vid : 0u], mb = off[(live ? vid : 0u) + 1u];
      u32 cnt = 0, mfirst = 0;
      if (live && P1 > P) {
        const u32 pad = P1 - P;
        const u32 x0 = lo > P ? lo - P : 0u, x1 = (hi < P1 ? hi : P1) - P;       // my part of the row's padded ranks
        if (x1 > x0) {
          const u32 len = mb - ma;
          const u32 c0 = (u32)((u64)x0 * len / pad), c1 = (u32)((u64)x1 * len / pad);
          mfirst = ma + c0; cnt = c1 - c0;
        }
      }
      const u32 incl = wave_inclusive_sum(cnt);
      const u32 T = (u32)__shfl((int)incl, WAVE - 1, WAVE);
      wave_lds_fence();
      pre[lane] = incl - cnt;
      m0s[lane] = mfirst;
      wave_lds_fence();
      // ---- the batch's mini-units, 64 at a time
      for (u32 t0 = 0; t0 < T; t0 += WAVE) {
        const u32 t = t0 + (u32)lane;
        const bool on = t < T;
        // the row of mini-unit t: the LAST r with pre[r] <= t (pre is the exclusive scan: non-decreasing; a row without mini-units
        // shares its value with the row behind it, so the last one at or below t is the row whose range holds t)
        u32 r = 0;
#pragma unroll
        for (u32 step = WAVE / 2; step > 0; step >>= 1) {
          const u32 cand = r + step;
          if (pre[cand] <= (on ? t : 0u)) r = cand;
        }
        const u32 idx = on ? m0s[r] + (t - pre[r]) : 0u;
        const nr_u32x4 d = __builtin_nontemporal_load((const nr_u32x4*)mu + idx);
        if (!tail) {
          const u32 o[8] = {d.x & 0xFFFFu, d.x >> 16, d.y & 0xFFFFu, d.y >> 16, d.z & 0xFFFFu, d.z >> 16, d.w & 0xFFFFu, d.w >> 16};
          u32 wd[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) wd[q] = bits[o[q] >> 5];
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            if (on && !((wd[q] >> (o[q] & 31u)) & 1u)) {
              (void)__hip_atomic_fetch_or(&bits[o[q] >> 5], 1u << (o[q] & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
              ++marks;
            }
          }
        } else {
          const u32 e[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            // (the cold-edge pass, when the slot runs it, owns what lies behind the LDS prefix of the other bodies)
            if (on && (int)e[q] >= 0 && !(cold_pass_runs && e[q] >= a.sliced_cold_from)) { mark[e[q]] = 1; ++marks; }
          }
        }
      }
      // ---- next batch: while rows that start below hi remain
      const u32 last_P = (u32)__shfl((int)(in_q ? P : 0xFFFFFFFFu), WAVE - 1, WAVE);
      seg += WAVE;
      if (seg >= nf || last_P == 0xFFFFFFFFu || last_P >= hi) break;
      wave_lds_fence();
    }
  }
  __syncthreads();
  if (!tail) {
    u32* const out = a.slice_flush + (size_t)block * BFS_SL_WORDS;
    for (u32 i = threadIdx.x; i < (u32)BFS_SL_WORDS; i += NT) out[i] = bits[i];
  }
  bfs_body_finish(a, marks, slot, stat_level, s_int);
}

}  // namespace mgx
