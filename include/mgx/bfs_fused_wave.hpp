// mgx/bfs_fused_wave.hpp -- push over the SHORT-row queue of a level: load-balanced search per edge rank,
// wave-private.
//
// Rows shorter than args.long_min cannot be streamed row-wise (a wave instruction would mostly idle), so
// their edges are addressed by RANK: the queue stores the exclusive degree scan next to every row, a wave
// owns a contiguous slice of the level's edge ranks, stages the (offset,row) pairs of the next 64 rows in its
// own LDS region and every lane resolves its ranks by a uniform-step binary search there (the reference's
// transform_lbs shape, advance.hxx:47-59, without the workgroup: cross-lane steps are ballots, and there is
// no barrier after the initial copy of the hot bitmap).  Measured on the earlier workgroup-synchronous
// kernels: 68 % of the cycles parked at 3-4 barriers per tile (rocprofv3 PMC, profiles/).
//
// Visited test: hot prefix of the bitmap in LDS (ds_or doubles as the exact intra-workgroup dedup); cold
// vertices probe the L2-resident bitmap word (COLDT) or are marked untested; 3-deep software pipeline with
// unconditional, countable loads (hipcc's s_waitcnt insertion is static: a load under a run-time condition
// drains the pipeline).  A neighbour that may be new gets mark[v] = 1, a plain byte store (bfs_fused.hpp);
// labels and the next level's queues are k_bfs_build's job.  Ranks are 32-bit throughout.
#pragma once
#include "bfs_fused.hpp"

namespace mgx {

constexpr int BFS_WAVE_EPT = 4;                       // ranks per lane per wave tile
constexpr int BFS_WAVE_TILE = WAVE * BFS_WAVE_EPT;    // 256 ranks
constexpr int BFS_WAVE_LDS_PER_WAVE = (68 + 64) * 4;  // staged (offset,row) pairs of 64 rows

constexpr size_t bfs_wave_lds_bytes(int nt, int hotw) {
  return (size_t)hotw * 4 + (size_t)(nt / 64) * BFS_WAVE_LDS_PER_WAVE + 64;
}

template <int NT, int HOTW, bool COLDT, bool NTLOAD = false>
__device__ __forceinline__ void bfs_wave_body(const bfs_fused_args_t& a, int level, u32 block, u32 nblocks, int stat_level) {
  constexpr int NW = NT / WAVE;
  constexpr int EPT = BFS_WAVE_EPT;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32* const hot = (u32*)smem;
  const int wave = threadIdx.x / WAVE;
  const int lane = lane_id();
  u32* const w_off = hot + HOTW + wave * (BFS_WAVE_LDS_PER_WAVE / 4);   // 65 used (+3 pad)
  u32* const w_row = w_off + 68;                                         // 64
  int* const s_int = (int*)(hot + HOTW + NW * (BFS_WAVE_LDS_PER_WAVE / 4));   // [0] marks stored

  bfs_ctrl_t* const c = a.ctrl;          // (`level` is the slot: ring and queue-buffer index, see bfs_resolve)
  const u64 cur = c->cursor[level % 3];
  const long long nf = (long long)(cur >> BFS_VSHIFT);
  const u32 E = (u32)(cur & BFS_EMASK);
  if (nf == 0 || bfs_level_pulls(a, c, level)) return;

  const u32* __restrict__ fr_row = a.fr_row[level & 1];
  const u32* __restrict__ fr_off = a.fr_off[level & 1];
  const u32* __restrict__ vis = a.visited;
  unsigned char* __restrict__ mark = a.mark;

  // slice of this wave
  const u32 total_waves = nblocks * NW;
  u32 per = (E + total_waves - 1) / total_waves;
  per = (per + WAVE - 1) / WAVE * WAVE;      // (not whole tiles: a level of a few thousand one-edge rows is spread over
                                             //  four times the waves, one tile of 64 rows each instead of four in a row)
  const u64 rb = (u64)(block * NW + wave) * per;
  const bool has_work = rb < (u64)E;
  const u32 r_begin = has_work ? (u32)rb : E;
  const u32 r_end = (rb + per < (u64)E) ? (u32)(rb + per) : E;

  // the hot prefix of the snapshot (grid-uniform decision: worth it only for big levels)
  const bool use_hot = E >= a.hot_min_edges;
  const u32 hot_n = use_hot ? (((u32)a.n < (u32)(HOTW * 32)) ? (u32)a.n : (u32)(HOTW * 32)) : 0u;
  if (use_hot) {
    bfs_copy_prefix<NT, HOTW>(hot, vis);
  }
  if (threadIdx.x == 0) s_int[0] = 0;
  __syncthreads();

  int marks = 0;       // per lane
  // marks of the vertices in [0, defer_n) wait for the end of the workgroup (bfs_hot_epilogue)
  const u32 defer_n = bfs_defer_limit(a, hot_n);

  if (has_work) {
    // first segment of the slice
    long long seg = wave_upper_bound(fr_off, nf, r_begin) - 1;

    // prefetch registers for the next tile's 64 (+1) segments: unconditional loads, validity applied later
    u32 pf_off = 0, pf_row = 0, pf_off_last = 0;
    bool pf_ok = false, pf_last_ok = false;
    auto prefetch = [&](long long sg) {
      const long long s0 = sg + lane;
      const long long s1 = sg + WAVE;
      pf_ok = s0 < nf;
      pf_last_ok = s1 < nf;
      pf_off = fr_off[pf_ok ? s0 : nf - 1];
      pf_row = fr_row[pf_ok ? s0 : nf - 1];
      pf_off_last = fr_off[pf_last_ok ? s1 : nf - 1];
    };
    prefetch(seg);

    u32 eidxC[EPT];
    u32 actC = 0;
    u32 e_next = r_begin;
    auto prepare_tile = [&]() {
      const u32 E0 = e_next;
      const u32 my = pf_ok ? pf_off : E;
      w_off[lane] = my;
      w_row[lane] = pf_ok ? pf_row : 0u;
      const u32 off64 = pf_last_ok ? pf_off_last : E;      // identical in every lane
      if (lane == 0) w_off[WAVE] = off64;
      u32 E1 = (r_end - E0 > (u32)BFS_WAVE_TILE) ? E0 + BFS_WAVE_TILE : r_end;
      if (off64 < E1) E1 = off64;                          // the 64 staged segments end here
      wave_lds_fence();                                    // the reads below are of OTHER lanes' slots (wave.hpp)
      const u32 nxt = w_off[lane + 1];
      const u64 m = __ballot(my < E1 && nxt >= E1);        // exactly one lane
      const int nseg = __ffsll((long long)m);              // 1..64
      const u32 next_off = w_off[nseg];
      const long long seg_next = seg + ((next_off == E1) ? nseg : nseg - 1);
      prefetch(seg_next);
      u32 r[EPT];
      int sj[EPT];
      actC = 0;
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const u32 rr = E0 + (u32)(k * WAVE + lane);
        const bool act = rr < E1;
        if (act) actC |= 1u << k;
        r[k] = act ? rr : E0;
        sj[k] = 0;
      }
      // The queue is built in vertex order (k_bfs_build) and the hub-first layout sorts vertices by degree, so the
      // 64 staged rows usually all have the SAME degree d: then rank x (relative to the first staged row) belongs
      // to row x / d -- one multiply-high with a per-tile reciprocal instead of a 6-step search in LDS.
      const u32 off0 = w_off[0];
      const u32 d0 = w_off[1] - off0;
      const bool uniform = d0 < 64u && __ballot(lane < nseg && (nxt - my) != d0) == 0ull;   // (x < 4096: the reciprocal is exact)
      if (uniform) {
        const u32 recip = d0 > 1u ? 0xFFFFFFFFu / d0 + 1u : 0u;         // ceil(2^32 / d0); x < 64 * 64: exact
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
          const u32 x = r[k] - off0;
          const u32 j = d0 > 1u ? __umulhi(x, recip) : x;
          eidxC[k] = w_row[j] + (x - j * d0);
        }
      } else {
        if (nseg > 1) {
          int top = 1;
          while (top * 2 < nseg) top *= 2;
          for (int step = top; step > 0; step >>= 1) {
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
              const int j = sj[k] + step;
              const u32 vv = w_off[j < nseg ? j : nseg - 1];
              if (j < nseg && vv <= r[k]) sj[k] = j;
            }
          }
        }
#pragma unroll
        for (int k = 0; k < EPT; ++k) eidxC[k] = w_row[sj[k]] + (r[k] - w_off[sj[k]]);
      }
      seg = seg_next;
      e_next = E1;
    };

    int dstA[EPT], dstB[EPT];
    u32 wordA[EPT];
    u32 actA = 0, actB = 0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) { dstA[k] = 0; dstB[k] = 0; wordA[k] = 0xFFFFFFFFu; }

    bool haveA = false, haveB = false, haveC = true;
    prepare_tile();
    while (haveA || haveB || haveC) {
      // ---- S4: tile it-2: hot neighbours claim their bit in the LDS copy (exact intra-workgroup dedup), cold
      //      ones test the bitmap word (COLDT) or are marked untested (k_bfs_build tests the bitmap anyway) -------
      if (haveA) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
          const u32 d = (u32)dstA[k];
          const u32 bit = 1u << (d & 31);
          if ((actA >> k) & 1u) {
            bool is_new;
            if (d < hot_n) is_new = !(hot[d >> 5] & bit) && !(atomicOr(&hot[d >> 5], bit) & bit);
            else is_new = COLDT ? !(wordA[k] & bit) : true;
            if (is_new) { if (d >= defer_n) mark[d] = 1; ++marks; }
          }
        }
      }
      // ---- S3a / S2 / S3b: unconditional, countable loads: a load under a run-time condition makes hipcc drain
      //      the pipeline at its next s_waitcnt ------------------------------------------------------------------
#pragma unroll
      for (int k = 0; k < EPT; ++k) dstA[k] = dstB[k];
      actA = haveB ? actB : 0u;
#pragma unroll
      for (int k = 0; k < EPT; ++k)
        dstB[k] = NTLOAD ? __builtin_nontemporal_load(a.col_indices + (haveC ? eidxC[k] : 0u)) : a.col_indices[haveC ? eidxC[k] : 0u];
      actB = haveC ? actC : 0u;
      if constexpr (COLDT) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) wordA[k] = vis[((u32)dstA[k] >= hot_n) ? ((u32)dstA[k] >> 5) : 0u];
      }
      haveA = haveB;
      haveB = haveC;
      // ---- S1: tile it+1 -------------------------------------------------------------------------------------
      haveC = e_next < r_end;
      if (haveC) prepare_tile();
    }
  }

  (void)bfs_hot_epilogue<NT>(a, hot, (defer_n + 31u) >> 5, level, s_int + 4, marks);
  bfs_body_finish(a, marks, level, stat_level, s_int);
}


}  // namespace mgx
