// mgx/bfs_fused_hot.hpp -- level kernel of the fused push BFS with the hot part of the visited
// snapshot resident in LDS.
//
// Why: with one visited probe per edge served by L2, a traversal cannot beat the L2 random-gather rate
// (measured 283 G gathers/s on MI355X => 0.47 ms for RMAT-22, tools/microbench.hip).  LDS serves
// random 4-byte reads an order of magnitude faster, and a skewed graph sends almost all of its edges
// to few vertices: in RMAT-22 the 2^19 highest-degree vertices (64 KB of bitmap) receive ~91 % of
// all edge endpoints.  So graphs are laid out hub-first (vertex ids sorted by descending degree,
// mini_amd.rmat.degree_order / mgx_graph_attach_layout) and every workgroup copies the first HOTW
// words of the level-start snapshot into LDS; only the cold ~9 % of the probes go to L2.
// The LDS copy doubles as an exact intra-workgroup dedup: staging a candidate sets its bit with a
// ds_or, later edges of the same workgroup to that vertex are dropped on the spot.
//
// Everything else is the design of bfs_fused.hpp (packed-cursor frontier append, LBS over the
// scanned frontier, snapshot + live bitmaps, batched claims, 3-deep software pipeline) with two
// simplifications: a tile ends where its NT staged segments end (no multi-round staging), and the
// workgroup size is a template parameter (512 threads, 2 workgroups per CU: one can flush while the
// other streams).
#pragma once
#include "bfs_fused.hpp"

namespace mgx {

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

constexpr int BFS_HOT_NT = 512;          // threads per workgroup
constexpr int BFS_HOT_EPT = 4;           // edge ranks per lane per tile
constexpr int BFS_HOT_WORDS = 16384;     // 64 KB of bitmap = 524288 vertices per workgroup
constexpr size_t bfs_hot_lds_bytes(int nt, int ept, int hotw) {
  // hot bitmap + s_off[nt+4] + s_row[nt] + st_v[nt + nt*ept/2] + scan u64[nt/64+1] + 16 misc u64
  return (size_t)hotw * 4 + (size_t)(nt + 4) * 4 + (size_t)nt * 4 + (size_t)(nt + nt * ept / 2) * 4 +
         (size_t)(nt / 64 + 1) * 8 + 16 * 8;
}

template <int NT, int EPT, int HOTW, bool DIAG>
__global__ __launch_bounds__(NT) void k_bfs_push_level_hot(bfs_fused_args_t a, int level) {
  constexpr int NW = NT / WAVE;
  constexpr int TILE = NT * EPT;
  constexpr int HALF_K = EPT / 2;
  constexpr int FLUSH_AT = NT;
  constexpr int STAGE = FLUSH_AT + NT * HALF_K;    // a half tile can add at most NT*HALF_K candidates
  constexpr int PER = STAGE / NT;
  constexpr u64 CNT1 = 1ull << 40;
  constexpr u64 DEGMASK = CNT1 - 1ull;
  static_assert(EPT % 2 == 0 && STAGE % NT == 0, "tile shape");

  // DIAG: s_memtime stamps WITHOUT forced waits (thread 0 of every workgroup): a section is charged
  // with whatever its first consumer had to wait for.  Sums land in ctrl->diag[]; never used for timing.
  long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long dt = 0;
#define MGX_HSTAMP(slot)                                                       \
  if (DIAG) {                                                                  \
    const long long now_ = (long long)__builtin_readcyclecounter();            \
    dg[slot] += now_ - dt;                                                     \
    dt = now_;                                                                 \
  }

  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32* const hot = (u32*)smem;
  u32* const s_off = hot + HOTW;            // NT + 1 (+3 pad)
  u32* const s_row = s_off + NT + 4;        // NT
  u32* const st_v = s_row + NT;             // STAGE
  u64* const s_scan = (u64*)(st_v + STAGE); // NW + 1
  u64* const s_misc = s_scan + NW + 1;      // [0] base, [1] seg, then ints
  int* const s_int = (int*)(s_misc + 2);    // [0] count [1] wins [2] nseg [3] claims

  bfs_ctrl_t* const c = a.ctrl;
  const u64 cur = c->cursor[level % 3];
  const long long nf = (long long)(cur >> BFS_VSHIFT);
  const u64 E = cur & BFS_EMASK;
  if (nf == 0 || c->pull || c->kind != 0) return;   // bookkeeping, direction and kernel choice: k_bfs_level_begin

  const u32* __restrict__ fr_row = a.fr_row[level & 1];
  const u32* __restrict__ fr_off = a.fr_off[level & 1];
  u32* __restrict__ out_row = a.fr_row[(level + 1) & 1];
  u32* __restrict__ out_off = a.fr_off[(level + 1) & 1];
  u64* const out_cursor = &c->cursor[(level + 1) % 3];

  u64 per = (E + gridDim.x - 1) / gridDim.x;
  per = (per + TILE - 1) / TILE * TILE;
  const u64 e_begin = (u64)blockIdx.x * per;
  if (e_begin >= E) return;
  const u64 e_end = (e_begin + per < E) ? e_begin + per : E;

  // hot bitmap: worth its 64 KB copy only when the workgroup streams a few tiles
  const bool use_hot = per >= (u64)a.hot_min_tiles * (u64)TILE;
  const u32 hot_n = use_hot ? (u32)(((u32)a.n < (u32)(HOTW * 32)) ? (u32)a.n : (u32)(HOTW * 32)) : 0u;
  if (use_hot) {
    const uint4* src = (const uint4*)a.snapshot;
    uint4* dstp = (uint4*)hot;
    for (int i = threadIdx.x; i < HOTW / 4; i += NT) dstp[i] = src[i];
  }
  if (threadIdx.x == 0) { s_int[0] = 0; s_int[1] = 0; s_int[3] = 0; }
  if (threadIdx.x < WAVE) {
    const long long ub = wave_upper_bound(fr_off, nf, (u32)e_begin);
    if (threadIdx.x == 0) s_misc[1] = (u64)(ub - 1);
  }
  __syncthreads();
  long long seg = (long long)s_misc[1];
  const int lane = lane_id();
  const int new_label = level + 1;
  const int* __restrict__ old_of_new = a.old_of_new;

  // ---- flush: batched claims + frontier append (see bfs_fused.hpp) ---------------------------------
  auto flush = [&](int cnt) {
    u32 v[PER], old[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = threadIdx.x * PER + q;
      v[q] = (i < cnt) ? st_v[i] : 0u;
      old[q] = (i < cnt) ? a.visited[v[q] >> 5] : 0xFFFFFFFFu;
    }
    u32 livemask = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q)
      if (!(old[q] & (1u << (v[q] & 31)))) livemask |= 1u << q;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(livemask) : : "memory");
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      old[q] = 0xFFFFFFFFu;
      if ((livemask >> q) & 1u) old[q] = atomicOr(a.visited + (v[q] >> 5), 1u << (v[q] & 31));
    }
    const int nclaim = wave_sum((int)__popc(livemask));
    if (lane == 0 && nclaim) atomicAdd(&s_int[3], nclaim);
    u32 winmask = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q)
      if (!(old[q] & (1u << (v[q] & 31)))) winmask |= 1u << q;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(winmask) : : "memory");
    if (a.append) {
    u32 ro[PER], ro1[PER];
    int lab_at[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const bool win = (winmask >> q) & 1u;
      const u32 w = win ? v[q] : 0u;
      ro[q] = a.row_offsets[w];
      ro1[q] = a.row_offsets[w + 1];
      lab_at[q] = old_of_new ? old_of_new[w] : (int)w;
    }
#pragma unroll
    for (int q = 0; q < PER; ++q)
      if ((winmask >> q) & 1u) a.labels[lab_at[q]] = new_label;
    u64 loc[PER];
    u64 sum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const u32 deg = ((winmask >> q) & 1u) ? ro1[q] - ro[q] : 0u;
      loc[q] = sum;
      sum += deg ? (CNT1 | (u64)deg) : 0ull;
    }
    u64 total;
    const u64 ex = block_exclusive_sum_nw<NW>(sum, s_scan, &total);
    const int nwin = wave_sum((int)__popc(winmask));
    if (lane == 0 && nwin) atomicAdd(&s_int[1], nwin);
    if (threadIdx.x == 0)
      s_misc[0] = (total >> 40) ? atomicAdd(out_cursor, ((total >> 40) << BFS_VSHIFT) | (total & DEGMASK)) : 0ull;
    __syncthreads();
    const u64 base = s_misc[0];
    const u64 base_v = base >> BFS_VSHIFT;
    const u64 base_e = base & BFS_EMASK;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      if (((winmask >> q) & 1u) && ro1[q] != ro[q]) {
        const u64 at = ex + loc[q];
        out_row[base_v + (at >> 40)] = ro[q];
        out_off[base_v + (at >> 40)] = (u32)(base_e + (at & DEGMASK));
      }
    }
    } else {
      const int nwin = wave_sum((int)__popc(winmask));
      if (lane == 0 && nwin) atomicAdd(&s_int[1], nwin);
    }
    if (threadIdx.x == 0) s_int[0] = 0;
    __syncthreads();
  };

  // ---- S1: stage NT segments, bound the tile by them, resolve (segment, rank) per edge rank ------------
  // Prefetch registers for the (offset,row) pairs of the coming tile.  The loads are UNCONDITIONAL
  // (index clamped, validity applied where the values are consumed): a load inside a branch makes
  // the number of outstanding loads unknowable to hipcc, which then drains everything with
  // s_waitcnt vmcnt(0) at the next use and serialises the software pipeline.
  u32 pf_off = 0, pf_row = 0, pf_off_last = 0;
  bool pf_ok = false, pf_last_ok = false;
  auto prefetch = [&](long long sg) {
    const long long s0 = sg + threadIdx.x;
    const long long s1 = sg + NT;                 // one extra offset so NT segments are usable
    pf_ok = s0 < nf;
    pf_last_ok = s1 < nf;
    pf_off = fr_off[pf_ok ? s0 : nf - 1];
    pf_row = fr_row[pf_ok ? s0 : nf - 1];
    pf_off_last = fr_off[pf_last_ok ? s1 : nf - 1];
  };
  prefetch(seg);
  u32 eidxC[EPT];
  u32 actC = 0;
  u64 e_next = e_begin;     // first rank not yet given to a tile
  auto prepare_tile = [&]() {
    const u64 E0 = e_next;
    s_off[threadIdx.x] = pf_ok ? pf_off : (u32)E;
    s_row[threadIdx.x] = pf_ok ? pf_row : 0u;
    if (threadIdx.x == 0) s_off[NT] = pf_last_ok ? pf_off_last : (u32)E;
    __syncthreads();
    u64 E1 = (E0 + TILE < e_end) ? E0 + TILE : e_end;
    if ((u64)s_off[NT] < E1) E1 = (u64)s_off[NT];      // the NT staged segments end here
    {
      const int j = threadIdx.x;                       // exactly one j: s_off[j] < E1 <= s_off[j+1]
      if ((u64)s_off[j] < E1 && (u64)s_off[j + 1] >= E1) s_int[2] = j + 1;
    }
    __syncthreads();
    const int nseg = s_int[2];
    const u32 next_off = s_off[nseg];
    const long long seg_next = seg + (((u64)next_off == E1) ? nseg : nseg - 1);
    if (E1 < e_end) prefetch(seg_next);
    u32 r32[EPT];
    int sj[EPT];
    actC = 0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const u64 r = E0 + (u64)(k * NT + threadIdx.x);
      const bool act = r < E1;
      if (act) actC |= 1u << k;
      r32[k] = act ? (u32)r : (u32)E0;
      sj[k] = 0;
    }
    if (nseg > 1) {
      int top = 1;
      while (top * 2 < nseg) top *= 2;
      for (int step = top; step > 0; step >>= 1) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
          const int j = sj[k] + step;
          const u32 v = s_off[j < nseg ? j : nseg - 1];
          if (j < nseg && v <= r32[k]) sj[k] = j;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) eidxC[k] = s_row[sj[k]] + (r32[k] - s_off[sj[k]]);
    seg = seg_next;
    e_next = E1;
  };

  // candidates of ranks k in [k0, k0+HALF_K): LDS dedup for hot ones, then stage
  int dstA[EPT], dstB[EPT];
  u32 wordA[EPT];
  u32 actA = 0, actB = 0;
#pragma unroll
  for (int k = 0; k < EPT; ++k) { dstA[k] = 0; dstB[k] = 0; wordA[k] = 0xFFFFFFFFu; }
  auto stage_half = [&](int k0) {
    u32 candmask = 0;
#pragma unroll
    for (int kk = 0; kk < HALF_K; ++kk) {
      const int k = k0 + kk;
      const u32 d = (u32)dstA[k];
      const u32 bit = 1u << (d & 31);
      if ((actA >> k) & 1u) {
        if (d < hot_n) {
          // the LDS copy is snapshot + this workgroup's own candidates: claim the bit locally
          if (!(hot[d >> 5] & bit) && !(atomicOr(&hot[d >> 5], bit) & bit)) candmask |= 1u << k;
        } else if (!(wordA[k] & bit)) {
          candmask |= 1u << k;
        }
      }
    }
    if (DIAG && (a.flags & 1) && level == (a.flags >> 8)) candmask = 0;
    // ballot compaction (no cross-lane scan: v_mbcnt ranks inside each ballot)
    u64 bal[HALF_K];
    int ncand = 0;
#pragma unroll
    for (int kk = 0; kk < HALF_K; ++kk) {
      bal[kk] = __ballot((candmask >> (k0 + kk)) & 1u);
      ncand += __popcll(bal[kk]);
    }
    if (ncand) {
      int base = 0;
      if (lane == 0) base = atomicAdd(&s_int[0], ncand);
      base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
      for (int kk = 0; kk < HALF_K; ++kk) {
        if ((candmask >> (k0 + kk)) & 1u) st_v[base + rank_in_mask(bal[kk])] = (u32)dstA[k0 + kk];
        base += __popcll(bal[kk]);
      }
    }
  };

  bool haveA = false, haveB = false, haveC = false;
  prepare_tile();
  haveC = true;
  if (DIAG) dt = (long long)__builtin_readcyclecounter();
  while (haveA || haveB || haveC) {
    // ---- S4 (first half): tile it-2 ------------------------------------------------------------------
    if (haveA) stage_half(0);
    MGX_HSTAMP(0)                        // wait for snapshot words + first half of the candidates
    __syncthreads();
    MGX_HSTAMP(1)                        // mid barrier
    {
      const int cnt = s_int[0];
      if (cnt >= FLUSH_AT) flush(cnt);
    }
    MGX_HSTAMP(2)                        // mid flush
    if (haveA) stage_half(HALF_K);
    MGX_HSTAMP(3)                        // second half of the candidates
    // The three load stages below are UNCONDITIONAL (validity travels in actA/actB/actC, addresses are
    // always safe): loads issued under a run-time condition make hipcc's vmcnt bookkeeping path
    // dependent, and it then drains the freshly issued loads at the next use of an older one.
    // ---- S3a: tile it-1: neighbour ids have landed: move them out of the col_indices landing registers
#pragma unroll
    for (int k = 0; k < EPT; ++k) dstA[k] = dstB[k];
    actA = haveB ? actB : 0u;
    // ---- S2: tile it: issue the col_indices reads (before the snapshot words: hipcc guards the reuse of
    //      the landing registers with an in-order vmcnt, which would otherwise wait for the words) ------
#pragma unroll
    for (int k = 0; k < EPT; ++k) dstB[k] = a.col_indices[haveC ? eidxC[k] : 0u];
    actB = haveC ? actC : 0u;
    // ---- S3b: tile it-1: cold vertices probe the L2-resident snapshot; hot lanes read word 0 (one
    //      broadcast request) so that exactly EPT loads are issued ----------------------------------------
#pragma unroll
    for (int k = 0; k < EPT; ++k)
      wordA[k] = a.snapshot[((u32)dstA[k] >= hot_n) ? ((u32)dstA[k] >> 5) : 0u];
    haveA = haveB;
    haveB = haveC;
    MGX_HSTAMP(4)                        // wait for col_indices + issue of the 2*EPT loads
    // ---- S1: tile it+1 ----------------------------------------------------------------------------------
    haveC = e_next < e_end;
    if (haveC) prepare_tile();
    MGX_HSTAMP(5)                        // staging + search of the next tile (2 barriers inside)
    __syncthreads();
    MGX_HSTAMP(6)                        // end barrier
    {
      const int cnt = s_int[0];
      if (cnt >= FLUSH_AT) flush(cnt);
    }
    MGX_HSTAMP(7)                        // end flush
  }
  {
    const int cnt = s_int[0];
    if (cnt > 0) flush(cnt);
  }
  if (threadIdx.x == 0 && s_int[1]) atomicAdd(&c->reached, (u64)s_int[1]);
  if (threadIdx.x == 0 && s_int[3]) {
    atomicAdd(&c->claims, (u64)s_int[3]);
    if (level < 64) atomicAdd(&c->claims_level[level], (u64)s_int[3]);
  }
  if (DIAG && threadIdx.x == 0 && (a.flags >> 8) == level)
    for (int i = 0; i < 8; ++i)
      if (dg[i]) atomicAdd(&c->diag[i], (u64)dg[i]);
#undef MGX_HSTAMP
}

// Runs a whole push BFS from `src` on the context's stream.  labels[] is (re)initialised here.
// Returns with the stream synchronised and host_ctrl holding the final counters.

// ---- bottom-up level (direction-optimising runs) ---------------------------------------------------
// One lane per vertex: an unvisited vertex walks its in-edges until it meets a member of the level's
// frontier bitmap (early exit -- the reference's advance_backward_kernel inspects every in-edge,
// advance.hxx:142-157).  A wave owns the two visited words of its 64 vertices, so discoveries are
// published with plain stores: no atomics at all in this direction.  Discovered vertices are also
// appended to the next frontier (packed cursor, as in the push kernel) so that sizes, termination
// and the TEPS numerator stay uniform across directions.
template <int NT>
__global__ __launch_bounds__(NT) void k_bfs_pull_level(bfs_fused_args_t a, int level) {
  constexpr int NW = NT / WAVE;
  constexpr int STAGE = 2 * NT;
  constexpr int PER = STAGE / NT;
  constexpr u64 CNT1 = 1ull << 40;
  constexpr u64 DEGMASK = CNT1 - 1ull;
  __shared__ u32 st_v[STAGE];
  __shared__ u64 s_scan[NW + 1];
  __shared__ u64 s_base;
  __shared__ unsigned long long s_insp;
  __shared__ int s_count, s_wins;

  bfs_ctrl_t* const c = a.ctrl;
  const u64 cur = c->cursor[level % 3];
  if ((cur >> BFS_VSHIFT) == 0 || !c->pull) return;
  u32* __restrict__ out_row = a.fr_row[(level + 1) & 1];
  u32* __restrict__ out_off = a.fr_off[(level + 1) & 1];
  u64* const out_cursor = &c->cursor[(level + 1) % 3];
  const int n = a.n;
  long long per_v = ((long long)n + gridDim.x - 1) / gridDim.x;
  per_v = (per_v + NT - 1) / NT * NT;
  const long long v_begin = (long long)blockIdx.x * per_v;
  if (v_begin >= n) return;
  const long long v_end = (v_begin + per_v < n) ? v_begin + per_v : n;
  if (threadIdx.x == 0) { s_count = 0; s_wins = 0; s_insp = 0ull; }
  __syncthreads();
  const int lane = lane_id();
  const int new_label = level + 1;
  const int* __restrict__ old_of_new = a.old_of_new;

  auto flush = [&](int cnt) {
    u32 v[PER], ro[PER], ro1[PER];
    int lab_at[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = threadIdx.x * PER + q;
      v[q] = (i < cnt) ? st_v[i] : 0u;
      ro[q] = a.row_offsets[v[q]];
      ro1[q] = a.row_offsets[v[q] + 1];
      lab_at[q] = old_of_new ? old_of_new[v[q]] : (int)v[q];
    }
    u64 loc[PER];
    u64 sum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = threadIdx.x * PER + q;
      if (i < cnt) a.labels[lab_at[q]] = new_label;
      const u32 deg = (i < cnt) ? ro1[q] - ro[q] : 0u;
      loc[q] = sum;
      sum += deg ? (CNT1 | (u64)deg) : 0ull;
    }
    u64 total;
    const u64 ex = block_exclusive_sum_nw<NW>(sum, s_scan, &total);
    if (threadIdx.x == 0) {
      s_base = (total >> 40) ? atomicAdd(out_cursor, ((total >> 40) << BFS_VSHIFT) | (total & DEGMASK)) : 0ull;
      s_wins += cnt;
    }
    __syncthreads();
    const u64 base = s_base;
    const u64 base_v = base >> BFS_VSHIFT;
    const u64 base_e = base & BFS_EMASK;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = threadIdx.x * PER + q;
      if (i < cnt && ro1[q] != ro[q]) {
        const u64 at = ex + loc[q];
        out_row[base_v + (at >> 40)] = ro[q];
        out_off[base_v + (at >> 40)] = (u32)(base_e + (at & DEGMASK));
      }
    }
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
  };

  for (long long base = v_begin; base < v_end; base += NT) {
    const long long v = base + threadIdx.x;
    const bool active = v < v_end;
    bool found = false;
    u32 word = 0xFFFFFFFFu;
    int inspected = 0;
    if (active) {
      word = a.snapshot[v >> 5];
      if (!((word >> (v & 31)) & 1u)) {
        const u32 e0 = a.in_offsets[v], e1 = a.in_offsets[v + 1];
        for (u32 e = e0; e < e1; ++e) {
          const u32 u = (u32)a.in_indices[e];
          ++inspected;
          if ((a.frontier_bits[u >> 5] >> (u & 31)) & 1u) { found = true; break; }
        }
      }
    }
    const u64 bal = __ballot(found);
    if (active && (lane & 31) == 0) {
      const u32 bits = (u32)(bal >> lane);
      if (bits) a.visited[v >> 5] = word | bits;     // this wave is the only writer of the word this level
    }
    const int nfound = __popcll(bal);
    if (nfound) {
      int at = 0;
      if (lane == 0) at = atomicAdd(&s_count, nfound);
      at = __builtin_amdgcn_readfirstlane(at);
      if (found) st_v[at + rank_in_mask(bal)] = (u32)v;
    }
    const int insp = wave_sum(inspected);
    if (lane == 0 && insp) atomicAdd(&s_insp, (unsigned long long)insp);
    __syncthreads();
    const int cnt = s_count;
    if (cnt >= NT) flush(cnt);
  }
  {
    const int cnt = s_count;
    if (cnt > 0) flush(cnt);
  }
  if (threadIdx.x == 0) {
    if (s_wins) atomicAdd(&c->reached, (u64)s_wins);
    if (s_insp) atomicAdd(&c->pull_edges, (u64)s_insp);
  }
}

}  // namespace mgx
