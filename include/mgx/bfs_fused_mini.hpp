// mgx/bfs_fused_mini.hpp -- a MID-SIZE level as one launch of a few workgroups (an "M" launch slot).
//
// A traversal of a skewed graph spends a fifth of its time on levels that hold a few thousand to a hundred thousand
// edges: too big for the one-workgroup chain (bfs_fused_chain.hpp: its claims are device-scope atomics from ONE compute
// unit, ~100 M/s, so it stops at a few thousand edges), tiny for a device-wide slot -- which costs them a push launch over
// 1152 workgroups plus a queue build that sweeps all n marks: ~21 us for the 4 500 edges of an RMAT-22 source's second
// level, ~21 us again for the 10 000 edges of the stragglers behind the peak.  Here such a level is ONE launch of
// BFS_MINI_WGS workgroups and needs no sweep at all:
//
//   * every workgroup copies the level's long-row queue (at most BFS_MINI_LCAP rows) into LDS and takes its share of the
//     64-edge units (the queue's offsets count padded degrees: one unit = one wave step, the row by a binary search in
//     LDS) and of the short rows (one row per thread);
//   * a neighbour whose bit reads unset is CLAIMED with atomicOr on the live bitmap -- exact, the winner is known at once
//     (64 compute units issue ~6 G claims/s: the level's discoveries are a few thousand);
//   * winners go to a list in LDS; the workgroup then reads their row extents, writes their labels and appends them to the
//     NEXT slot's two queues with one block scan and one packed cursor atomic per queue (k_bfs_build's scheme) -- at most
//     BFS_MINI_WGS atomics on the hot cursor instead of 512.
// The launch takes the place of a slot's [push, build] pair: it consumes the level of slot s and leaves the next one in
// slot s + 1's queues, ring entry and slot_level.  Which slots of a batch are M launches is the host's guess from the
// level sizes of the previous traversal of the graph (bfs_fused_run.hpp).  When the guess is wrong -- the level is too
// big (or lazy: no queues; or the traversal is direction-optimising) -- the launch FORWARDS the level unchanged to slot
// s + 1 (queues copied, sizes and flags moved): correct whatever the host guessed, at the price of the copy.
#pragma once
#include "bfs_fused.hpp"
#include "bfs_fused_chain.hpp"

namespace mgx {

constexpr int BFS_MINI_WGS = 64;               // workgroups of an M launch
constexpr int BFS_MINI_LCAP = 4096;            // long rows of a level an M launch expands (their queue is staged in LDS)
constexpr int BFS_MINI_WCAP = 8192;            // winners a workgroup collects before it flushes them
constexpr u32 BFS_MINI_EDGES_LATE = 131072;    // largest level (true edges) behind the peak ...
constexpr u32 BFS_MINI_EDGES_EARLY = 32768;    // ... and before it, where nearly every edge is a claim (64 CUs issue ~6 G claims/s; with 131072 here the
                                               // launch took the 114 000 all-new edges of some RMAT-22 sources' second level: slower than a slot)
constexpr u32 BFS_MINI_SHORT_ROWS = 65536;     // short rows of such a level
constexpr size_t bfs_mini_lds_bytes() {
  return (size_t)(2 * BFS_MINI_LCAP + 4) * 4 + (size_t)BFS_MINI_WCAP * 4 + 64 * 8 + 256;
}

// grid-uniform: may the level in ring entry slot % 3 be expanded by an M launch?
__device__ __forceinline__ bool bfs_level_is_mini(const bfs_fused_args_t& a, const bfs_ctrl_t* c, int slot) {
  if (a.mode != 0 || c->lazy_slot == slot) return false;
  const u64 cur = c->cursor[slot % 3], lcur = c->lcursor[slot % 3];
  const u64 E = (cur & BFS_EMASK) + c->ledges[slot % 3];
  const bool late = c->reached * 4ull >= (u64)(u32)a.n;
  return (lcur >> BFS_VSHIFT) <= (u64)BFS_MINI_LCAP && (cur >> BFS_VSHIFT) <= (u64)BFS_MINI_SHORT_ROWS &&
         E <= (u64)(late ? BFS_MINI_EDGES_LATE : BFS_MINI_EDGES_EARLY);
}

template <int NT>
__global__ __launch_bounds__(NT) void k_bfs_mini(bfs_fused_args_t a, int arg) {
  constexpr int NW = NT / WAVE;
  constexpr u64 CNT1 = 1ull << 40;
  constexpr u64 DEGMASK = CNT1 - 1ull;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32* const s_loff = (u32*)smem;                        // BFS_MINI_LCAP + 1 (+ pad)
  u32* const s_lrow = s_loff + BFS_MINI_LCAP + 4;        // BFS_MINI_LCAP
  u32* const s_win = s_lrow + BFS_MINI_LCAP;             // BFS_MINI_WCAP
  u64* const s_scan = (u64*)(s_win + BFS_MINI_WCAP);     // NW + 1 (<= 64)
  __shared__ int s_cnt, s_maxdeg;
  __shared__ u64 s_base[2];
  __shared__ u32 s_long_true;
  bfs_ctrl_t* const c = a.ctrl;
  int slot, level;
  bfs_resolve(c, arg, slot, level);
  if (c->done) return;
  const u64 cur = c->cursor[slot % 3], lcur = c->lcursor[slot % 3];
  const u64 ledges = c->ledges[slot % 3];
  const int nf_s = (int)(cur >> BFS_VSHIFT), nf_l = (int)(lcur >> BFS_VSHIFT);
  const u32 Es = (u32)(cur & BFS_EMASK), Rl = (u32)(lcur & BFS_EMASK);
  const int in = slot & 1, out = (slot + 1) & 1;
  const bool first_thread = blockIdx.x == 0 && threadIdx.x == 0;
  // NOTHING a workgroup bases its decisions on below (ring entry slot % 3, lazy_slot, reached, done, mode) is written by
  // this launch before every workgroup has made them: the bookkeeping touches the NEXT slot's entries only, discoveries
  // count into reached_mini, and a forwarded level's flags move when the last workgroup is through.
  if (nf_s + nf_l == 0) {                                // an empty frontier: the traversal is over
    if (first_thread) {
      c->cursor[(slot + 2) % 3] = 0; c->lcursor[(slot + 2) % 3] = 0; c->ledges[(slot + 2) % 3] = 0;
      if (level < 64) c->stamp[level] = __builtin_amdgcn_s_memrealtime();
      if (!c->done) { c->done = 1; c->levels = level; }
    }
    return;
  }
  const long long gtid = (long long)blockIdx.x * NT + threadIdx.x, gthreads = (long long)gridDim.x * NT;
  if (!bfs_level_is_mini(a, c, slot)) {
    // ---- forward: the level moves to slot + 1 as it is ------------------------------------------------------------
    const bool lazy = c->lazy_slot == slot;
    const bool fb = c->fb_slot == slot;
    if (!lazy) {
      for (long long i = gtid; i < nf_s; i += gthreads) { a.fr_row[out][i] = a.fr_row[in][i]; a.fr_off[out][i] = a.fr_off[in][i]; }
      for (long long i = gtid; i < nf_l; i += gthreads) { a.lq_row[out][i] = a.lq_row[in][i]; a.lq_off[out][i] = a.lq_off[in][i]; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      if (atomicAdd(&c->mini_blocks[slot & 3], 1u) == gridDim.x - 1u) {       // the last workgroup through: everybody has decided (and copied)
        c->mini_blocks[slot & 3] = 0u;
        c->cursor[(slot + 1) % 3] = cur; c->lcursor[(slot + 1) % 3] = lcur; c->ledges[(slot + 1) % 3] = ledges;
        c->cursor[(slot + 2) % 3] = 0; c->lcursor[(slot + 2) % 3] = 0; c->ledges[(slot + 2) % 3] = 0;
        c->slot_level[(slot + 1) & 3] = level;
        c->flush_count[(slot + 1) & 1] = 0;
        bfs_slot_marks_clear(a, slot + 1);
        if (lazy) c->lazy_slot = slot + 1;
        if (fb) c->fb_slot = slot + 1;
      }
    }
    return;
  }
  // ---- the level's bookkeeping (bfs_slot_open / bfs_open_level of a device-wide slot) -------------------------------
  if (first_thread) {
    const u64 E = (u64)Es + ledges;
    c->cursor[(slot + 2) % 3] = 0; c->lcursor[(slot + 2) % 3] = 0; c->ledges[(slot + 2) % 3] = 0;
    if (level < 64) c->stamp[level] = __builtin_amdgcn_s_memrealtime();
    if (level < BFS_MAX_TRACE) c->trace[level] = ((u64)(nf_s + nf_l) << BFS_VSHIFT) | E;
    c->sum_edges += E;
    c->sum_frontier += (u64)(nf_s + nf_l);
    c->sum_long_edges += ledges;
    c->sum_long_vertices += (u64)nf_l;
    c->push_levels += 1;
    c->small_levels += 1;
    c->mini_slots += 1;
    c->slots += 1;
    c->slot_level[(slot + 1) & 3] = level + 1;
    c->flush_count[(slot + 1) & 1] = 0;
    bfs_slot_marks_clear(a, slot + 1);
    c->fb_slot = -1;                       // (no frontier bitmap is written: the next level walks its queues)
  }
  const int lane = lane_id();
  const int wave = threadIdx.x / WAVE;
  const u32 long_min = a.long_min > 0 ? (u32)a.long_min : 0xFFFFFFFFu;
  const int new_label = level + 1;
  // stage the long-row queue
  for (int i = threadIdx.x; i < nf_l; i += NT) { s_loff[i] = a.lq_off[in][i]; s_lrow[i] = a.lq_row[in][i]; }
  if (threadIdx.x == 0) { s_loff[nf_l] = Rl; s_cnt = 0; s_maxdeg = 0; }
  __syncthreads();

  u64* const cur_s = &c->cursor[(slot + 1) % 3];
  u64* const cur_l = &c->lcursor[(slot + 1) % 3];
  u32* __restrict__ const out_row_s = a.fr_row[out];
  u32* __restrict__ const out_off_s = a.fr_off[out];
  u32* __restrict__ const out_row_l = a.lq_row[out];
  u32* __restrict__ const out_off_l = a.lq_off[out];
  int found = 0;                                           // winners this thread claimed (for ctrl->reached)

  // winners in s_win[0, s_cnt) -> labels, row extents, the next slot's queues (all threads; barriers inside)
  auto flush = [&]() {
    __syncthreads();
    const int W = s_cnt;
    for (int first = 0; first < W; first += NT) {          // (block-uniform)
      const int i = first + (int)threadIdx.x;
      u32 ro = 0, deg = 0;
      if (i < W) {
        const u32 v = s_win[i];
        const bfs_u32x2 ext = *(const bfs_u32x2*)(a.row_offsets + v);
        ro = ext.x; deg = ext.y - ext.x;
        a.labels[a.old_of_new ? a.old_of_new[v] : (int)v] = new_label;
      }
      const bool is_long = deg >= long_min;
      const u64 add_s = (!is_long && deg) ? (CNT1 | (u64)deg) : 0ull;
      const u64 add_l = is_long ? (CNT1 | (u64)bfs_lq_pad(deg)) : 0ull;
      if (threadIdx.x == 0) s_long_true = 0;
      u64 tot_s, tot_l;
      const u64 ex_s = block_exclusive_sum_lean<NW>(add_s, s_scan, &tot_s);      // (also orders s_long_true = 0 before the adds)
      const u64 ex_l = block_exclusive_sum_lean<NW>(add_l, s_scan, &tot_l);
      const u32 lt = wave_sum(is_long ? deg : 0u);
      if (lane == 0 && lt) atomicAdd(&s_long_true, lt);
      __syncthreads();
      if (threadIdx.x == 0) {
        s_base[0] = (tot_s >> 40) ? atomicAdd(cur_s, ((tot_s >> 40) << BFS_VSHIFT) | (tot_s & DEGMASK)) : 0ull;
        s_base[1] = (tot_l >> 40) ? atomicAdd(cur_l, ((tot_l >> 40) << BFS_VSHIFT) | (tot_l & DEGMASK)) : 0ull;
        if (tot_l >> 40) atomicAdd(&c->ledges[(slot + 1) % 3], (u64)s_long_true);
      }
      __syncthreads();
      if (deg) {
        const u64 b = is_long ? s_base[1] : s_base[0];
        const u64 at = is_long ? ex_l : ex_s;
        const u64 pos = (b >> BFS_VSHIFT) + (at >> 40);
        (is_long ? out_row_l : out_row_s)[pos] = ro;
        (is_long ? out_off_l : out_off_s)[pos] = (u32)((b & BFS_EMASK) + (at & DEGMASK)) | (is_long ? (deg & 63u) : 0u);
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
  };
  // one neighbour: test, claim, remember
  auto visit = [&](bool act, u32 d) {
    bool win = false;
    if (act) {
      const u32 bit = 1u << (d & 31u);
      const u32 seen = a.visited[d >> 5];
      if (!(seen & bit)) win = !(atomicOr(a.visited + (d >> 5), bit) & bit);
    }
    const u64 bal = __ballot(win);
    if (bal) {
      int at = 0;
      if (lane == 0) at = atomicAdd(&s_cnt, __popcll(bal));
      at = __builtin_amdgcn_readfirstlane(at);
      if (win) { s_win[at + rank_in_mask(bal)] = d; ++found; }
    }
  };

  // ---- long rows: 64-edge units (the queue's offsets count padded degrees), equal shares per wave -----------------------
  {
    const u32 U = Rl >> 6;
    const u32 Wt = (u32)gridDim.x * NW, w0 = (u32)blockIdx.x * NW + (u32)wave;
    const u32 iters = (U + Wt - 1u) / Wt;                  // (grid-uniform)
    for (u32 t = 0; t < iters; ++t) {
      const u32 u = w0 + t * Wt;
      bool act = false;
      u32 d = 0;
      if (u < U) {                                         // (wave-uniform)
        const u32 r0 = u << 6;
        int lo = 0, hi = nf_l;                             // last row with (s_loff[row] & ~63) <= r0
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if ((s_loff[mid] & ~63u) <= r0) lo = mid; else hi = mid; }
        const u32 e0 = s_loff[lo];
        const u32 deg = bfs_lq_degree(e0, s_loff[lo + 1]);
        const u32 rank = r0 - (e0 & ~63u) + (u32)lane;
        act = rank < deg;
        if (act) d = (u32)a.col_indices[s_lrow[lo] + rank];
      }
      visit(act, d);
      __syncthreads();
      // ONE snapshot of the count decides for the whole workgroup: without the second barrier a fast wave could pass the test,
      // run into the next visit() and push s_cnt over the limit before a slow wave has looked -- the waves would then disagree
      // about entering flush(), whose barriers and block scans must pair
      const int cnt_now = s_cnt;
      __syncthreads();
      if (cnt_now > BFS_MINI_WCAP - NT) flush();
    }
  }
  // ---- short rows, by EDGE RANK when they average four entries or more (round 6): a thread takes rank r of the queue's scanned
  // degrees, finds its row by bisection (the offsets of a mid-size level live in the L2; neighbouring lanes ask for neighbouring
  // rows) and visits ONE neighbour -- every edge of the level in one or two steps.  The row-per-thread walk below costs a row of 31
  // entries 31 dependent steps (neighbour, bitmap word, returning claim, two barriers each): a uniform random graph's third
  // level -- 981 rows, 32 114 entries, every one a discovery -- took 182 us that way.
  if (nf_s > 0 && (u64)Es >= 4ull * (u64)nf_s) {
    const u32* __restrict__ fr_row = a.fr_row[in];
    const u32* __restrict__ fr_off = a.fr_off[in];
    const u32 iters = (Es + (u32)gthreads - 1u) / (u32)gthreads;             // (grid-uniform)
    for (u32 t = 0; t < iters; ++t) {
      const u64 r64 = (u64)gtid + (u64)t * (u64)gthreads;
      const bool act = r64 < (u64)Es;
      const u32 r = act ? (u32)r64 : 0u;
      int lo = 0, hi = nf_s;                                 // last row with fr_off[row] <= r (fr_off[0] == 0)
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (fr_off[mid] <= r) lo = mid; else hi = mid; }
      const u32 d = act ? (u32)a.col_indices[fr_row[lo] + (r - fr_off[lo])] : 0u;
      visit(act, d);
      __syncthreads();
      const int cnt_now = s_cnt;                             // (one snapshot, as above)
      __syncthreads();
      if (cnt_now > BFS_MINI_WCAP - NT) flush();
    }
  } else
  // ---- short rows: one row per thread, edge k of every row in step k -----------------------------------------------------
  {
    const u32* __restrict__ fr_row = a.fr_row[in];
    const u32* __restrict__ fr_off = a.fr_off[in];
    const u32 iters = ((u32)nf_s + (u32)gthreads - 1u) / (u32)gthreads;     // (grid-uniform)
    for (u32 t = 0; t < iters; ++t) {
      const long long i = gtid + (long long)t * gthreads;
      u32 row = 0, deg = 0;
      if (i < nf_s) {
        const u32 e0 = fr_off[i], e1 = (i + 1 < nf_s) ? fr_off[i + 1] : Es;
        row = fr_row[i];
        deg = e1 - e0;
      }
      int wmax = (int)deg;
#pragma unroll
      for (int sh = WAVE / 2; sh > 0; sh >>= 1) { const int o = __shfl_xor(wmax, sh, WAVE); wmax = o > wmax ? o : wmax; }
      if (lane == 0 && wmax) atomicMax(&s_maxdeg, wmax);
      __syncthreads();
      const u32 kmax = (u32)s_maxdeg;
      for (u32 k = 0; k < kmax; ++k) {                      // (block-uniform)
        const bool act = k < deg;
        const u32 d = act ? (u32)a.col_indices[row + k] : 0u;
        visit(act, d);
        __syncthreads();
        const int cnt_now = s_cnt;                          // (one snapshot, as above)
        __syncthreads();
        if (cnt_now > BFS_MINI_WCAP - NT) flush();
      }
      __syncthreads();
      if (threadIdx.x == 0) s_maxdeg = 0;
      __syncthreads();
    }
  }
  flush();
  found = (int)wave_sum((u32)found);
  if (lane == 0 && found) atomicAdd(&c->reached_mini, (u64)found);
}

}  // namespace mgx
