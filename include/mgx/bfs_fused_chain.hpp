// mgx/bfs_fused_chain.hpp -- the levels that are too small for a device-wide pass, run by ONE workgroup of the push
// launch, back to back.
//
// A traversal of a skewed graph has 2-3 levels that carry the work and 4-5 that hold a handful of vertices (the source's
// neighbourhood at the start, the stragglers at the end).  Device-wide, such a level costs a push launch over 1024
// workgroups and a sweep over n marks whatever its size: ~20 us for 30 edges, and RMAT-22 has four of them (~80 of
// 450 us).  Here the push launch looks at the level first (every workgroup reads the same stable sizes,
// bfs_level_is_chained): a level of at most BFS_CHAIN_CAP edges is expanded by block 0 alone, which then keeps going with
// the levels behind it as long as they are small too -- no launch, no sweep and no cross-workgroup hand-off between them
// (one workgroup sees its own writes).  The other workgroups return at once.
//
//   stage in   the slot's two global queues -> ONE list in LDS: (row start, exclusive scan of TRUE degrees)
//   per level  bookkeeping (one thread); load-balanced search per edge rank in LDS; visited test + atomicOr claim on the
//              live bitmap (a few thousand device atomics at most: cheap at this size, and the winner is known at once);
//              winners -> labels, row extents, workgroup scan -> the next level's list, again in LDS
//   stage out  when the next level is big (or the chain limit is reached): its list -> the two global queues of the next
//              slot (long rows with offsets in padded units, bfs_lq_*), cursors, slot_level[s + 1]; skip_build[s] tells
//              the slot's k_bfs_build that there is nothing to sweep.
//
// What the other workgroups of the launch (which may start late) base their decision on -- ring entry s % 3, slot_level[s]
// -- is never written here: the chain keeps its sizes in LDS and publishes only into the NEXT slot's entries.
// Direction-optimising runs keep every level device-wide (the bottom-up kernel needs the frontier bitmap of k_bfs_build).
#pragma once
#include "bfs_fused.hpp"

namespace mgx {

constexpr int BFS_CHAIN_NT = 1024;
constexpr int BFS_CHAIN_CAP = 6144;          // edges (>= winners >= rows of the next level) of a chained level
constexpr int BFS_CHAIN_EPT = 4;             // edge ranks per thread in flight
constexpr int BFS_CHAIN_CAP_BIG = 12288;     // ... of a level run by the in-place chain kernel (a launch of its own: 148 KB of LDS)
constexpr size_t bfs_chain_lds_bytes(int cap = BFS_CHAIN_CAP) {
  return (size_t)(BFS_CHAIN_NT / 64 + 2) * 8 + (size_t)(3 * cap + 4) * 4 + 64;
}

// grid-uniform: sizes of ring entry slot % 3 (complete since the previous launch)
// What a chained level may cost.  The claims are device-scope atomics from ONE compute unit, which issues about 100 M of
// them per second (tools/microbench.hip: 25 G/s over 256 CUs): a level that discovers 4 000 vertices takes 50 us here and
// 20 us device-wide (measured: 335 edges 14 us, 1 937 26 us, 4 532 55 us, 7 440 74 us for the first two levels of a
// traversal).  The claim is only issued for a neighbour whose bit reads unset, so what counts is the DISCOVERIES: early
// in a traversal nearly every edge is one (limit BFS_CHAIN_EARLY_EDGES edges); once a quarter of the vertices is reached
// the peak is over and the stragglers' edges mostly end at visited vertices (limit: max_edges).
constexpr u32 BFS_CHAIN_EARLY_EDGES = 1536;
// the direction rule of bfs_level_pulls for a frontier of nf vertices with `reached` vertices labelled
__device__ __forceinline__ bool bfs_rule_pulls(const bfs_fused_args_t& a, u64 reached, u64 nf) {
  const float unvisited = (float)((long long)a.n - (long long)reached);
  return unvisited < (float)(long long)nf * a.alpha;
}
__device__ __forceinline__ u32 bfs_chain_edge_limit(const bfs_fused_args_t& a, u64 reached, u32 max_edges) {
  const bool late = reached * 4ull >= (u64)(u32)a.n;
  return late || max_edges < BFS_CHAIN_EARLY_EDGES ? max_edges : BFS_CHAIN_EARLY_EDGES;
}
__device__ __forceinline__ bool bfs_level_is_chained(const bfs_fused_args_t& a, u64 cur, u64 lcur, u64 ledges, u32 max_edges, int list_cap = BFS_CHAIN_CAP) {
  if (max_edges == 0u) return false;
  const u64 nf = (cur >> BFS_VSHIFT) + (lcur >> BFS_VSHIFT);
  const u64 E = (cur & BFS_EMASK) + ledges;
  // direction-optimising runs: only top-down levels are chained (the rule of bfs_level_pulls, same inputs)
  if (a.mode == 1 && (a.ctrl->pull || bfs_rule_pulls(a, a.ctrl->reached, nf))) return false;
  const u32 lim = bfs_chain_edge_limit(a, a.ctrl->reached, max_edges);
  const u64 cap = lim < (u32)list_cap ? lim : (u32)list_cap;
  return nf <= (u64)list_cap && E <= cap;
}
__device__ __forceinline__ bool bfs_level_is_chained(const bfs_fused_args_t& a, u64 cur, u64 lcur, u64 ledges) {
  return bfs_level_is_chained(a, cur, lcur, ledges, a.chain_max_edges);
}

// Runs level `level` of slot `slot` and the small levels behind it.  Whole workgroup (NT threads).
// INPLACE: the chain is a launch of its own in FRONT of the slot (k_bfs_chain_inplace, bfs_fused_run.hpp): it takes the
// slot's queues and leaves the first level that is not small in the SAME slot's queues and ring entry -- the slot's push
// launch then opens that level; nothing is skipped and no slot is used up.
template <int NT, bool INPLACE = false, int CAP = BFS_CHAIN_CAP>
__device__ __forceinline__ void bfs_chain_body(const bfs_fused_args_t& a, int slot, int level) {
  constexpr int NW = NT / WAVE;
  constexpr int EPT = BFS_CHAIN_EPT;
  constexpr u64 CNT1 = 1ull << 40;
  constexpr u64 DEGMASK = CNT1 - 1ull;
  static_assert(CAP % NT == 0, "chain list shape");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u64* const s_scan = (u64*)smem;                      // NW + 1
  u32* const s_off = (u32*)(s_scan + NW + 2);          // CAP + 2: exclusive scan of true degrees, s_off[nf] = E
  u32* const s_row = s_off + CAP + 2;                  // CAP
  u32* const s_win = s_row + CAP;                      // CAP: vertices claimed in the level being expanded
  int* const s_i = (int*)(s_win + CAP);                // [0] winners
  bfs_ctrl_t* const c = a.ctrl;
  const int lane = lane_id();
  const u32 long_min = a.long_min > 0 ? (u32)a.long_min : 0xFFFFFFFFu;
  const u32 lim_e = INPLACE ? a.chain_big_edges : a.chain_max_edges;

  // ---- stage in: the slot's two queues as one list (long rows first) ---------------------------------------------
  int nf;
  u32 E;
  {
    const u64 cur = c->cursor[slot % 3], lcur = c->lcursor[slot % 3];
    const int nf_l = (int)(lcur >> BFS_VSHIFT), nf_s = (int)(cur >> BFS_VSHIFT);
    const u32 El = (u32)(lcur & BFS_EMASK), Es = (u32)(cur & BFS_EMASK);
    const u32* __restrict__ lq_row = a.lq_row[slot & 1];
    const u32* __restrict__ lq_off = a.lq_off[slot & 1];
    const u32* __restrict__ fr_row = a.fr_row[slot & 1];
    const u32* __restrict__ fr_off = a.fr_off[slot & 1];
    nf = nf_l + nf_s;
    u64 run = 0;                       // edges of the list so far (rounds of NT entries: no per-thread arrays, this
                                       // body shares its launch's 64-register budget with the streaming bodies)
    for (int first = 0; first < nf; first += NT) {
      const int i = first + threadIdx.x;
      u32 row = 0, deg = 0;
      if (i < nf_l) {
        const u32 e0 = lq_off[i], e1 = (i + 1 < nf_l) ? lq_off[i + 1] : El;
        row = lq_row[i];
        deg = bfs_lq_degree(e0, e1);
      } else if (i < nf) {
        const int j = i - nf_l;
        const u32 e0 = fr_off[j], e1 = (j + 1 < nf_s) ? fr_off[j + 1] : Es;
        row = fr_row[j];
        deg = e1 - e0;
      }
      u64 tot;
      const u64 ex = run + block_exclusive_sum_lean<NW>((u64)deg, s_scan, &tot);
      if (i < nf) { s_row[i] = row; s_off[i] = (u32)ex; }
      run += tot;
    }
    E = (u32)run;
    if (threadIdx.x == 0) s_off[nf] = E;
  }

  for (int chained = 0;; ++chained) {
    if (threadIdx.x == 0) {
      // the level's bookkeeping (what bfs_open_level does for a device-wide level)
      if (level < 64) c->stamp[level] = __builtin_amdgcn_s_memrealtime();
      if (level < BFS_MAX_TRACE) c->trace[level] = ((u64)nf << BFS_VSHIFT) | (u64)E;
      c->sum_edges += (u64)E;
      c->sum_frontier += (u64)nf;
      c->push_levels += 1;
      c->small_levels += 1;
      if (chained == 0 && !INPLACE) c->slots += 1;
      s_i[0] = 0;
    }
    __syncthreads();

    // ---- expand: load-balanced search per edge rank, claim on the live bitmap --------------------------------------
    {
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
              if (j < nf && s_off[j] <= r[k]) sj[k] = j;
            }
          }
        int d[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) d[k] = a.col_indices[act[k] ? s_row[sj[k]] + (r[k] - s_off[sj[k]]) : 0u];
        // a look at the word first (possibly stale, i.e. with fewer bits: then the claim below decides): one compute unit
        // issues ~100 M device-scope atomics per second, and the stragglers' edges mostly end at visited vertices
        u32 seen[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) seen[k] = act[k] ? a.visited[(u32)d[k] >> 5] : 0xFFFFFFFFu;
        u32 old[EPT];
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
          const u32 bit = 1u << (d[k] & 31);
          old[k] = (seen[k] & bit) ? 0xFFFFFFFFu : atomicOr(a.visited + ((u32)d[k] >> 5), bit);
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
    }
    __syncthreads();                    // winners complete; the list in s_off / s_row is dead from here

    // ---- winners -> labels, the next level's list (in place of the old one) ---------------------------------------
    const int W = s_i[0];               // <= E <= CAP
    const int new_label = level + 1;
    u64 run = 0;                        // (count << 40 | edges) of the list so far
    constexpr int WPT = 4;              // winners per thread and round (4096 per round: one round for most chained levels)
    for (int first = 0; first < W; first += WPT * NT) {
      u32 ro[WPT], dg[WPT];
      u64 mine = 0;
#pragma unroll
      for (int q = 0; q < WPT; ++q) {
        const int i = first + threadIdx.x * WPT + q;
        ro[q] = 0; dg[q] = 0;
        if (i < W) {
          const u32 v = s_win[i];
          const bfs_u32x2 ext = *(const bfs_u32x2*)(a.row_offsets + v);
          ro[q] = ext.x;
          dg[q] = ext.y - ext.x;
          a.labels[a.old_of_new ? a.old_of_new[v] : (int)v] = new_label;
        }
        if (dg[q]) mine += CNT1 | (u64)dg[q];
      }
      u64 tot;
      u64 ex = run + block_exclusive_sum_lean<NW>(mine, s_scan, &tot);
#pragma unroll
      for (int q = 0; q < WPT; ++q) {
        if (dg[q]) {
          s_row[ex >> 40] = ro[q];
          s_off[ex >> 40] = (u32)(ex & DEGMASK);
          ex += CNT1 | (u64)dg[q];
        }
      }
      run += tot;
    }
    const int nf2 = (int)(run >> 40);
    const u32 E2 = (u32)(run & DEGMASK);          // (a level reached from <= CAP edges: far below 2^32)
    if (threadIdx.x == 0) {
      s_off[nf2] = E2;
      const u64 reached_now = c->reached + (u64)W;
      c->reached = reached_now;
      s_i[1] = (int)bfs_chain_edge_limit(a, reached_now, lim_e);      // what the next level may hold to be chained too
      s_i[2] = (a.mode == 1 && bfs_rule_pulls(a, reached_now, (u64)nf2)) ? 1 : 0;   // direction-optimising: the next level is a bottom-up one
      if (a.count_marks) { c->claims += (u64)W; if (level < 64) c->claims_level[level] += (u64)W; }
    }
    __syncthreads();

    const u32 max_l = (u32)s_i[1];
    const u32 max_e = max_l < (u32)CAP ? max_l : (u32)CAP;
    const bool next_pulls = s_i[2] != 0;
    if (nf2 != 0 && (run & DEGMASK) <= (u64)max_e && !next_pulls) {     // the next level is small too (and top-down): keep going
      nf = nf2;
      E = E2;
      level += 1;
      continue;
    }

    // ---- stage out: the next level's list -> the global queues of slot + 1 (in place: of this slot) ----------------------------------------
    const int out = INPLACE ? slot : slot + 1;      // (in place: the list is in LDS, the queues it came from are dead)
    u32* __restrict__ const out_row_s = a.fr_row[out & 1];
    u32* __restrict__ const out_off_s = a.fr_off[out & 1];
    u32* __restrict__ const out_row_l = a.lq_row[out & 1];
    u32* __restrict__ const out_off_l = a.lq_off[out & 1];
    u64 tot_s = 0, tot_l = 0, tot_true = 0;        // (count << 40 | edges) of the two queues so far; true edges of the long one
    for (int first = 0; first < nf2; first += NT) {
      const int i = first + threadIdx.x;
      u32 ro = 0, dg = 0;
      if (i < nf2) { ro = s_row[i]; dg = s_off[i + 1] - s_off[i]; }
      const bool is_long = dg >= long_min;
      const u64 add_s = (!is_long && dg) ? (CNT1 | (u64)dg) : 0ull;
      const u64 add_l = is_long ? (CNT1 | (u64)bfs_lq_pad(dg)) : 0ull;
      u64 ts, tl, tt;
      const u64 ex_s = tot_s + block_exclusive_sum_lean<NW>(add_s, s_scan, &ts);
      const u64 ex_l = tot_l + block_exclusive_sum_lean<NW>(add_l, s_scan, &tl);
      (void)block_exclusive_sum_lean<NW>(is_long ? (u64)dg : 0ull, s_scan, &tt);
      if (is_long) {
        out_row_l[ex_l >> 40] = ro;
        out_off_l[ex_l >> 40] = (u32)(ex_l & DEGMASK) | (dg & 63u);
      } else if (dg) {
        out_row_s[ex_s >> 40] = ro;
        out_off_s[ex_s >> 40] = (u32)(ex_s & DEGMASK);
      }
      tot_s += ts; tot_l += tl; tot_true += tt;
    }
    // ... and, when that level will read its long rows from the unit blocks (bfs_long_is_dense's rule), its frontier as
    // a bitmap -- what k_bfs_build leaves behind a device-wide level: s_win still holds the vertices this level discovered.
    if (nf2 != 0 && (next_pulls || (a.ub_owner && a.dense_div && ((tot_l & DEGMASK) >> 6) * (u64)a.dense_div >= (u64)a.ub_units))) {
      uint4* const fb4 = (uint4*)a.frontier_bits;
      const int quads = (a.n + 127) / 128;
      for (int i = threadIdx.x; i < quads; i += NT) fb4[i] = make_uint4(0u, 0u, 0u, 0u);
      __threadfence();                 // (the atomics below must land on cleared words: same workgroup, through L2)
      __syncthreads();
      for (int i = threadIdx.x; i < W; i += NT) {
        const u32 v = s_win[i];
        atomicOr(a.frontier_bits + (v >> 5), 1u << (v & 31u));
      }
      if (threadIdx.x == 0) c->fb_slot = out;
    } else if (INPLACE) {
      if (threadIdx.x == 0) c->fb_slot = -1;       // (the init kernel's bitmap -- the source alone -- is not this slot's frontier any more)
    }
    if (threadIdx.x == 0) {
      c->cursor[out % 3] = ((tot_s >> 40) << BFS_VSHIFT) | (tot_s & DEGMASK);
      c->lcursor[out % 3] = ((tot_l >> 40) << BFS_VSHIFT) | (tot_l & DEGMASK);
      c->ledges[out % 3] = tot_true;
      c->slot_level[out & 3] = level + 1;
      if (!INPLACE) {
        c->cursor[(slot + 2) % 3] = 0;
        c->lcursor[(slot + 2) % 3] = 0;
        c->ledges[(slot + 2) % 3] = 0;
        c->skip_build[slot & 3] = 1;
        c->flush_count[(slot + 1) & 1] = 0;
        bfs_slot_marks_clear(a, slot + 1);
      }
      if (nf2 == 0 && !c->done) { c->done = 1; c->levels = level + 1; }
    }
    return;
  }
}

}  // namespace mgx
