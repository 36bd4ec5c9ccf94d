// mgx/bfs_fused_small.hpp -- the levels that are too small for a device-wide launch.
//
// A traversal of a skewed graph has 2-3 levels that carry the work and 4-5 that hold a handful of vertices (the
// source's neighbourhood at the start, the stragglers at the end).  A device-wide level costs four launches
// (~5 us each back to back, measured) plus a sweep over n marks, whatever its size: ~35 us for 30 edges.
// Here ONE workgroup runs such levels back to back inside one launch:
//
//   loop:  open the level (bookkeeping, bfs_open_level)
//          frontier empty               -> done
//          more than SMALL_MAX_E edges  -> leave it to the device-wide kernels of this slot (ctrl->big = 1), return
//          else expand it right here: both queues copied to LDS, load-balanced search per edge rank in LDS,
//          visited test + atomicOr claim on the live bitmap (a few thousand device atomics at most: cheap at this
//          size, and the winner is known at once), winners staged in LDS; then labels, row extents, a workgroup
//          scan, and the next level's two queues written directly (nobody else appends: no cursor atomics).
//
// The host launches slots of [k_bfs_small_levels, stream, wave, (pull,) build] with level = -1: the device-wide
// kernels read ctrl->level and return at once unless this kernel opened a big level for them.  The next slot's
// first action is to close that level.  Direction-optimising runs keep every level on the device-wide path (the
// bottom-up kernel needs the frontier bitmap that k_bfs_build maintains).
#pragma once
#include "bfs_fused.hpp"

namespace mgx {

constexpr int BFS_SMALL_NT = 1024;
constexpr int BFS_SMALL_MAX_E = 8192;       // edges of a level the single workgroup expands itself
constexpr int BFS_SMALL_EPT = 4;            // edge ranks per thread in flight
constexpr size_t bfs_small_lds_bytes() {
  return (size_t)(BFS_SMALL_NT / 64 + 1 + 2) * 8 + (size_t)(3 * BFS_SMALL_MAX_E + 2) * 4 + 16;
}

template <int NT>
__global__ __launch_bounds__(NT) void k_bfs_small_levels(bfs_fused_args_t a, u32 small_max_e) {
  constexpr int NW = NT / WAVE;
  constexpr int CAP = BFS_SMALL_MAX_E;
  constexpr int EPT = BFS_SMALL_EPT;
  constexpr int PER = CAP / NT;
  constexpr u64 CNT1 = 1ull << 40;
  constexpr u64 DEGMASK = CNT1 - 1ull;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u64* const s_scan = (u64*)smem;                      // NW + 1
  u64* const s_cur = s_scan + NW + 1;                  // [0] long-row cursor [1] short-row cursor of the level
  u32* const s_off = (u32*)(s_cur + 2);                // CAP + 1 (+1 pad): exclusive degree scan of the queue being expanded
  u32* const s_row = s_off + CAP + 2;                  // CAP
  u32* const s_win = s_row + CAP;                      // CAP: vertices claimed in this level
  int* const s_i = (int*)(s_win + CAP);                // [0] winners [1] state: 0 expand, 1 big, 2 done [2] level
  bfs_ctrl_t* const c = a.ctrl;
  const int lane = lane_id();
  if (small_max_e > (u32)CAP) small_max_e = (u32)CAP;

  if (threadIdx.x == 0) {
    if (c->big) { c->level += 1; c->big = 0; }           // the previous slot's big level is finished
    if (!c->done) c->slots += 1;
  }
  for (;;) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const int level = c->level;
      int state = 0;
      if (c->done || !bfs_open_level(a, level)) state = 2;
      else {
        const u64 E = (c->cursor[level % 3] & BFS_EMASK) + (c->lcursor[level % 3] & BFS_EMASK);
        if (a.mode == 1 || E > (u64)small_max_e) { c->big = 1; state = 1; }
        else c->small_levels += 1;
      }
      s_i[0] = 0; s_i[1] = state; s_i[2] = level;
      s_cur[0] = c->lcursor[level % 3];
      s_cur[1] = c->cursor[level % 3];
    }
    __syncthreads();
    if (s_i[1] != 0) return;
    const int level = s_i[2];

    // ---- expand both queues of the level (long rows first) ------------------------------------------------------
    for (int which = 0; which < 2; ++which) {
      const u64 cur = s_cur[which];
      const int nf = (int)(cur >> BFS_VSHIFT);
      const u32 E = (u32)(cur & BFS_EMASK);
      if (nf == 0) continue;
      // volatile: these arrays were written by this workgroup one level ago and read two levels ago -- the loads
      // must not be served from a stale L1 line
      const volatile u32* q_row = (which ? a.fr_row : a.lq_row)[level & 1];
      const volatile u32* q_off = (which ? a.fr_off : a.lq_off)[level & 1];
      for (int i = threadIdx.x; i < nf; i += NT) { s_off[i] = q_off[i]; s_row[i] = q_row[i]; }
      if (threadIdx.x == 0) s_off[nf] = E;
      // long-row queue: padded offsets with degree & 63 in the low bits (bfs_lq_* in bfs_fused.hpp)
      const u32 omask = which ? 0xFFFFFFFFu : ~63u;
      __syncthreads();
      int top = 1;
      while (top * 2 < nf) top *= 2;
      for (u32 base = 0; base < E; base += NT * EPT) {
        u32 r[EPT];
        int sj[EPT];
        bool act[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
          r[k] = base + (u32)(k * NT) + threadIdx.x;
          act[k] = r[k] < E;
          if (!act[k]) r[k] = 0;
          sj[k] = 0;
        }
        if (nf > 1)
          for (int step = top; step > 0; step >>= 1) {
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
              const int j = sj[k] + step;
              if (j < nf && (s_off[j] & omask) <= r[k]) sj[k] = j;
            }
          }
        int d[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
          const u32 e0 = s_off[sj[k]];
          const u32 in_row = r[k] - (e0 & omask);
          if (!which) act[k] = act[k] && in_row < bfs_lq_degree(e0, s_off[sj[k] + 1]);     // ranks in a row's padding
          d[k] = a.col_indices[act[k] ? s_row[sj[k]] + in_row : 0u];
        }
        u32 word[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) word[k] = a.visited[(u32)d[k] >> 5];
        u32 old[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
          const u32 bit = 1u << (d[k] & 31);
          old[k] = 0xFFFFFFFFu;
          if (act[k] && !(word[k] & bit)) old[k] = atomicOr(a.visited + ((u32)d[k] >> 5), bit);
        }
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
          const bool win = !(old[k] & (1u << (d[k] & 31)));
          const u64 bal = __ballot(win);
          if (bal) {
            int at = 0;
            if (lane == 0) at = atomicAdd(&s_i[0], __popcll(bal));
            at = __builtin_amdgcn_readfirstlane(at);
            if (win) s_win[at + rank_in_mask(bal)] = (u32)d[k];
          }
        }
      }
      __syncthreads();         // s_off / s_row are reused by the other queue
    }

    // ---- winners -> labels and the queues of level + 1 ------------------------------------------------------------
    const int W = s_i[0];      // <= edges of the level <= CAP
    const int new_label = level + 1;
    const u32 long_min = a.long_min > 0 ? (u32)a.long_min : 0xFFFFFFFFu;
    u32 ro[PER], ro1[PER];
    int lab_at[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = threadIdx.x * PER + q;
      const u32 v = (i < W) ? s_win[i] : 0u;
      ro[q] = a.row_offsets[v];
      ro1[q] = a.row_offsets[v + 1];
      lab_at[q] = a.old_of_new ? a.old_of_new[v] : (int)v;
    }
    u64 loc[PER];
    u64 sum_s = 0, sum_l = 0, true_l = 0;
    u32 longmask = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = threadIdx.x * PER + q;
      if (i < W) a.labels[lab_at[q]] = new_label;
      const u32 deg = (i < W) ? ro1[q] - ro[q] : 0u;
      const bool is_long = deg >= long_min;
      if (is_long) { longmask |= 1u << q; true_l += deg; }
      loc[q] = is_long ? sum_l : sum_s;
      const u64 add = deg ? (CNT1 | (u64)(is_long ? bfs_lq_pad(deg) : deg)) : 0ull;
      if (is_long) sum_l += add; else sum_s += add;
    }
    u64 tot_s, tot_l, tot_true;
    const u64 ex_s = block_exclusive_sum_nw<NW>(sum_s, s_scan, &tot_s);
    const u64 ex_l = block_exclusive_sum_nw<NW>(sum_l, s_scan, &tot_l);
    (void)block_exclusive_sum_nw<NW>(true_l, s_scan, &tot_true);
    u32* __restrict__ const out_row_s = a.fr_row[(level + 1) & 1];
    u32* __restrict__ const out_off_s = a.fr_off[(level + 1) & 1];
    u32* __restrict__ const out_row_l = a.lq_row[(level + 1) & 1];
    u32* __restrict__ const out_off_l = a.lq_off[(level + 1) & 1];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = threadIdx.x * PER + q;
      if (i < W && ro1[q] != ro[q]) {
        const bool is_long = (longmask >> q) & 1u;
        const u64 at = (is_long ? ex_l : ex_s) + loc[q];
        (is_long ? out_row_l : out_row_s)[at >> 40] = ro[q];
        (is_long ? out_off_l : out_off_s)[at >> 40] = (u32)(at & DEGMASK) | (is_long ? ((ro1[q] - ro[q]) & 63u) : 0u);
      }
    }
    if (threadIdx.x == 0) {
      c->cursor[(level + 1) % 3] = ((tot_s >> 40) << BFS_VSHIFT) | (tot_s & DEGMASK);
      c->lcursor[(level + 1) % 3] = ((tot_l >> 40) << BFS_VSHIFT) | (tot_l & DEGMASK);
      c->ledges[(level + 1) % 3] = tot_true;
      c->reached += (u64)W;
      c->claims += (u64)W;
      if (level < 64) c->claims_level[level] += (u64)W;
      c->level = level + 1;
    }
    // the queue stores above must be visible to this workgroup's next iteration (same CU: through L2 is enough)
    __threadfence();
  }
}

}  // namespace mgx
