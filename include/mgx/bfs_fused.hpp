// mgx/bfs_fused.hpp -- device-resident push BFS: advance + filter fused per level, no host
// round trip per level.
//
// What the reference does per level: degree scan (K1) -> 4-byte D2H (K2) ->
// load-balanced expand writing one int per EDGE, mostly -1 (K3) -> compaction upsweep + D2H (K4)
// -> downsweep (K5); ~10 launches, 2 host syncs, 4-5 cudaMalloc/cudaFree pairs, and
// 16 B/edge + 40 B/vertex of traffic.  Here one kernel per level does all of it:
//
//   * the frontier is stored as (row_start, scanned_edge_offset) pairs.  The scan that the
//     reference recomputes each level is produced for free when a level APPENDS its
//     discoveries: a workgroup flushes its LDS-staged discoveries with ONE 64-bit atomicAdd on
//     a packed (vertex_count << 38 | edge_count) cursor, so the slot it gets back is at once
//     the frontier position and the exclusive degree scan of that position;
//   * every workgroup owns a contiguous slice of the level's edge ranks [0,E): one
//     wave-cooperative 64-ary search finds its first segment, after which it streams:
//     per tile of BFS_TILE edges the (offset,row) slice is staged in LDS (coalesced) and each
//     lane resolves its edges by binary search in LDS -- lanes of a wave read consecutive
//     col_indices of a row;
//   * visited test = one bit per vertex (n/8 bytes: L2-resident), claim = atomicOr on that
//     word, winner stores the label and stages (row_start, degree) of the new vertex;
//     zero-degree discoveries are labelled but never enter the frontier;
//   * termination, per-level sizes and the TEPS numerator stay on the device; the host reads
//     one flag every `levels_per_sync` launches.
//
// Algorithmic traffic: 8 B per traversed edge (col index + visited/label probe) and 20 B per
// frontier vertex -- the figure BASELINE.md's roofline uses.
#pragma once
#include "runtime.hpp"
#include "wave.hpp"

namespace mgx {

constexpr int BFS_FLUSH_AT = 512;              // staged discoveries that trigger a flush
constexpr int BFS_MAX_TRACE = 4096;            // per-level trace slots
constexpr int BFS_VSHIFT = 38;                 // cursor = (vertices << 38) | edges
constexpr u64 BFS_EMASK = (1ull << BFS_VSHIFT) - 1ull;

struct bfs_ctrl_t {
  u64 cursor[3];     // level L reads [L%3], appends into [(L+1)%3], clears [(L+2)%3]
  u64 merged_new;    // partitioned BFS (bfs_dist2.hpp): vertices discovered by ALL ranks in the level just merged
  u64 sum_edges;     // sum over levels of E  == m_t (out-degrees of reached vertices)
  u64 sum_frontier;  // sum over levels of frontier sizes (reached vertices with degree >= 1)
  u64 reached;       // vertices labelled (incl. source and zero-degree discoveries)
  u64 claims;        // atomicOr claims issued (>= reached-1; the excess is lost races / stale reads)
  u64 claims_level[64];
  u64 diag[8];       // DIAG builds only: cycles per stage, summed over workgroups (thread 0 stamps)
  int done;
  int levels;        // number of levels that expanded at least one edge
  int pull;          // direction of the level about to run (set by k_bfs_level_begin; sticky once 1)
  int push_levels;   // levels run top-down
  u64 kind_mask;     // bit L set: level L (< 64) ran on the wave-private streaming kernel
  int kind;          // top-down kernel for the level about to run: 0 = workgroup-synchronous tiles
                     // (discovery-heavy levels: big flushes), 1 = wave-private streaming (k_bfs_level_begin)
  u64 pull_edges;    // in-edges inspected by bottom-up levels
  u64 trace[BFS_MAX_TRACE];   // cursor value each level started from (kept LAST: read back only up to `levels`)
};

struct bfs_fused_args_t {
  const u32* row_offsets;
  const int* col_indices;
  int* labels;
  u32* visited;        // (n+31)/32 words: claimed with atomicOr (memory-side: every atomic drops its L2 line)
  const u32* snapshot; // copy of `visited` taken between levels: read-only while a level runs, stays L2-resident
  u32* fr_row[2];      // frontier: CSR row start of each frontier vertex
  u32* fr_off[2];      // frontier: exclusive scan of degrees (edge rank of its first edge)
  bfs_ctrl_t* ctrl;
  u32* frontier_bits;      // direction-optimising runs: bitmap of the level's frontier (visited now & ~snapshot before)
  const u32* in_offsets;   // in-edges for bottom-up levels (== row_offsets/col_indices on symmetric graphs)
  const int* in_indices;
  int wave_kernel;         // 1: levels whose average frontier degree is below wave_max_avg_degree use the wave kernel
  int wave_max_avg_degree;
  int append;              // 1: winners are appended to the next frontier (single GPU); 0: claims only -- the
                           // partitioned BFS rebuilds every rank's frontier from the exchanged bitmaps
  int mode;                // MGX_BFS_PUSH / MGX_BFS_DIRECTION_OPT
  float alpha;             // switch to bottom-up when unvisited < frontier_vertices * alpha (bfs_enactor.hxx:68)
  const int* old_of_new;   // hub-first layout: original id of layout vertex v (NULL = identity)
  const int* new_of_old;   // and its inverse
  int n;
  int hot_min_tiles;   // hot-bitmap kernel: LDS copy is used when a workgroup streams at least this many tiles
  int flags;           // diagnostics only: bit 0 = skip the claims (results are then wrong by design)
};

__global__ void k_bfs_fused_init(bfs_fused_args_t a, int src) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  bfs_ctrl_t* c = a.ctrl;
  a.labels[src] = 0;                                   // labels live in ORIGINAL id space
  if (a.new_of_old) src = a.new_of_old[src];           // everything else in layout space
  a.visited[src >> 5] = 1u << (src & 31);
  const u32 ro = a.row_offsets[src];
  const u32 deg = a.row_offsets[src + 1] - ro;
  a.fr_row[0][0] = ro;
  a.fr_off[0][0] = 0;
  c->cursor[0] = deg ? ((1ull << BFS_VSHIFT) | (u64)deg) : 0ull;
  c->cursor[1] = 0;
  c->cursor[2] = 0;
  c->sum_edges = 0;
  c->sum_frontier = 0;
  c->reached = 1;
  c->claims = 0;
  for (int i = 0; i < 64; ++i) c->claims_level[i] = 0;
  for (int i = 0; i < 8; ++i) c->diag[i] = 0;
  c->done = 0;
  c->levels = 0;
  c->pull = 0;
  c->push_levels = 0;
  c->kind = 0;
  c->kind_mask = 0;
  c->pull_edges = 0;
}

// Runs before the kernel(s) of every level: per-level bookkeeping (termination flag, trace, TEPS
// numerator), the direction decision, and the level-start snapshot of the visited bitmap --
// plus, for direction-optimising runs, the frontier bitmap (what became visited since the last snapshot).
__global__ __launch_bounds__(BLOCK) void k_bfs_level_begin(bfs_fused_args_t a, int level, long long nwords) {
  bfs_ctrl_t* const c = a.ctrl;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    const u64 cur = c->cursor[level % 3];
    const long long nf = (long long)(cur >> BFS_VSHIFT);
    const u64 E = cur & BFS_EMASK;
    c->cursor[(level + 2) % 3] = 0;
    if (nf == 0) {
      if (!c->done) { c->done = 1; c->levels = level; }
    } else {
      if (level < BFS_MAX_TRACE) c->trace[level] = cur;
      c->sum_edges += E;
      c->sum_frontier += (u64)nf;
      if (a.mode == 1 && !c->pull) {
        const float unvisited = (float)((long long)a.n - (long long)c->reached);
        if (unvisited < (float)nf * a.alpha) c->pull = 1;      // bfs_enactor.hxx:68; never switches back (:74-112)
      }
      if (!c->pull) c->push_levels += 1;
      // long rows (a hub frontier) discover a lot per edge: batch the claims per workgroup; short rows
      // (the big levels of a skewed graph) mostly hit visited vertices: stream them wave by wave
      c->kind = (a.wave_kernel && E / (u64)nf < (u64)a.wave_max_avg_degree) ? 1 : 0;
      if (c->kind == 1 && !c->pull && level < 64) c->kind_mask |= 1ull << level;
    }
  }
  const bool want_frontier = a.mode == 1;
  const u32* __restrict__ vis = a.visited;
  u32* __restrict__ snap = (u32*)a.snapshot;
  for (long long w = (long long)blockIdx.x * BLOCK + threadIdx.x; w < nwords; w += (long long)gridDim.x * BLOCK) {
    const u32 now = vis[w];
    if (want_frontier) a.frontier_bits[w] = now & ~snap[w];
    snap[w] = now;
  }
}

// EPT = edge ranks per lane per tile (tile = BLOCK*EPT ranks, strided by the workgroup so a wave's
// 64 lanes read 64 consecutive col_indices).  The per-edge dependency chain is
//   LDS search -> col_indices load -> visited-word load -> atomicOr claim
// and every stage is issued for all EPT ranks before the next stage starts, so a lane keeps EPT
// independent memory operations in flight.  Two round trips are kept OFF the per-tile chain:
//   * the (offset,row) slice of the next tile is prefetched into registers while this tile runs;
//   * winners are staged in LDS as bare vertex ids; their row extents are fetched, scanned and
//     appended to the next frontier in batches (flush), one round trip per ~BFS_FLUSH_AT winners.
// DIAG builds stamp s_memtime at the stage boundaries (thread 0 of every workgroup) and add the
// per-stage cycle totals to ctrl->diag[]; they are never used for timing claims.
template <int EPT, int OCC, bool DIAG = false>
__global__ __launch_bounds__(BLOCK, OCC) void k_bfs_push_level(bfs_fused_args_t a, int level) {
  long long dg[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  long long dt = 0;
#define MGX_STAMP(slot)                                                        \
  if (DIAG) {                                                                  \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                \
    const long long now_ = (long long)__builtin_readcyclecounter();            \
    dg[slot] += now_ - dt;                                                     \
    dt = now_;                                                                 \
  }

  constexpr int TILE = BLOCK * EPT;
  constexpr int STAGE = BFS_FLUSH_AT + TILE;       // multiple of BLOCK
  constexpr int PER = STAGE / BLOCK;
  constexpr u64 CNT1 = 1ull << 40;                 // flush scan item = (kept << 40) | degree
  constexpr u64 DEGMASK = CNT1 - 1ull;
  __shared__ u32 s_off[TILE + 1];
  __shared__ u32 s_row[TILE];
  __shared__ u32 st_v[STAGE];
  __shared__ u64 s_scan[WAVES_PER_BLOCK + 1];
  __shared__ u64 s_base;
  __shared__ long long s_seg;
  __shared__ int s_count, s_wins, s_nseg, s_claims;

  bfs_ctrl_t* const c = a.ctrl;
  const u64 cur = c->cursor[level % 3];
  const long long nf = (long long)(cur >> BFS_VSHIFT);
  const u64 E = cur & BFS_EMASK;
  if (nf == 0 || c->pull) return;   // bookkeeping and direction: k_bfs_level_begin

  const u32* __restrict__ fr_row = a.fr_row[level & 1];
  const u32* __restrict__ fr_off = a.fr_off[level & 1];
  u32* __restrict__ out_row = a.fr_row[(level + 1) & 1];
  u32* __restrict__ out_off = a.fr_off[(level + 1) & 1];
  u64* const out_cursor = &c->cursor[(level + 1) % 3];

  // this workgroup's slice of edge ranks, tile aligned
  u64 per = (E + gridDim.x - 1) / gridDim.x;
  per = (per + TILE - 1) / TILE * TILE;
  const u64 e_begin = (u64)blockIdx.x * per;
  if (e_begin >= E) return;
  const u64 e_end = (e_begin + per < E) ? e_begin + per : E;

  if (threadIdx.x == 0) { s_count = 0; s_wins = 0; s_claims = 0; }
  if (threadIdx.x < WAVE) {
    const long long ub = wave_upper_bound(fr_off, nf, (u32)e_begin);
    if (threadIdx.x == 0) s_seg = ub - 1;
  }
  __syncthreads();
  long long seg = s_seg;
  const int lane = lane_id();
  const int new_label = level + 1;

  // Flush the staged CANDIDATES (vertices whose visited bit read as clear): claim them with one
  // atomicOr each -- all in flight together, ONE round trip for the whole batch instead of one
  // per tile (a returning atomic executes at the memory side, ~5 us under load, and stalls the
  // whole wave: measured 0.8 ms of a 1.3 ms level for 1 % of the edges) -- then winners get
  // their label, their row extent is fetched, zero-degree ones are dropped and the rest is
  // appended to the next frontier.  ONE packed 64-bit atomicAdd hands back the frontier slot
  // AND the exclusive degree scan at that slot.
  auto flush = [&](int cnt) {
    u32 v[PER], old[PER];
    // (1) the staged vertices were unvisited in the level-start snapshot; most duplicates (same
    //     vertex reached again later in this level) are weeded out by a plain read of the LIVE
    //     bitmap, (2) the rest is claimed.
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
    if (lane == 0 && nclaim) atomicAdd(&s_claims, nclaim);
    u32 winmask = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q)
      if (!(old[q] & (1u << (v[q] & 31)))) winmask |= 1u << q;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(winmask) : : "memory");
    u32 ro[PER], ro1[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const bool win = (winmask >> q) & 1u;
      const u32 w = win ? v[q] : 0u;
      ro[q] = a.row_offsets[w];
      ro1[q] = a.row_offsets[w + 1];
      if (win) a.labels[a.old_of_new ? a.old_of_new[v[q]] : (int)v[q]] = new_label;
    }
    u64 loc[PER];
    u64 sum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const u32 deg = ((winmask >> q) & 1u) ? ro1[q] - ro[q] : 0u;
      loc[q] = sum;
      sum += deg ? (CNT1 | (u64)deg) : 0ull;
    }
    u64 total;
    const u64 ex = block_exclusive_sum(sum, s_scan, &total);
    const int nwin = wave_sum((int)__popc(winmask));
    if (lane == 0 && nwin) atomicAdd(&s_wins, nwin);
    if (threadIdx.x == 0)
      s_base = (total >> 40) ? atomicAdd(out_cursor, ((total >> 40) << BFS_VSHIFT) | (total & DEGMASK)) : 0ull;
    __syncthreads();
    const u64 base = s_base;
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
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
  };

  // prefetch registers for the first BLOCK+1 (offset,row) pairs of the coming tile
  // Prefetch registers for the (offset,row) pairs of the coming tile.  The loads are UNCONDITIONAL
  // (index clamped, validity applied where the values are consumed): a load inside a branch makes
  // the number of outstanding loads unknowable to hipcc, which then drains everything with
  // s_waitcnt vmcnt(0) at the next use and serialises the software pipeline.
  u32 pf_off = 0, pf_row = 0, pf_off_last = 0;
  bool pf_ok = false, pf_last_ok = false;
  auto prefetch = [&](long long sg) {
    const long long s0 = sg + threadIdx.x;
    const long long s1 = sg + BLOCK;                 // one extra offset so BLOCK segments are usable
    pf_ok = s0 < nf;
    pf_last_ok = s1 < nf;
    pf_off = fr_off[pf_ok ? s0 : nf - 1];
    pf_row = fr_row[pf_ok ? s0 : nf - 1];
    pf_off_last = fr_off[pf_last_ok ? s1 : nf - 1];
  };
  prefetch(seg);

  // S1: resolve the CSR position of every edge rank of the tile starting at E0 (LDS only, apart
  // from extra staging rounds for tiles spanning more than BLOCK segments).
  u32 eidxC[EPT];
  u32 actC = 0;
  auto prepare_tile = [&](u64 E0) {
    const u64 E1 = (E0 + TILE < e_end) ? E0 + TILE : e_end;
    // stage (scanned offset, row start) of the segments beginning at `seg`: the prefetched round
    // first, more rounds of BLOCK only if the tile spans more than BLOCK segments.  Segments are
    // non-empty, so TILE+1 offsets always suffice.
    s_off[threadIdx.x] = pf_ok ? pf_off : (u32)E;
    s_row[threadIdx.x] = pf_ok ? pf_row : 0u;
    if (threadIdx.x == 0) s_off[BLOCK] = pf_last_ok ? pf_off_last : (u32)E;
    __syncthreads();
    int limit = BLOCK;   // index of the last staged offset
    while ((u64)s_off[limit] < E1 && limit < TILE) {
      const int j = limit + 1 + threadIdx.x;
      if (j <= TILE) {
        const long long s = seg + j;
        s_off[j] = (s < nf) ? fr_off[s] : (u32)E;
      }
      const int jr = limit + threadIdx.x;
      if (jr < TILE) {
        const long long s = seg + jr;
        s_row[jr] = (s < nf) ? fr_row[s] : 0u;
      }
      limit = (limit + BLOCK < TILE) ? limit + BLOCK : TILE;
      __syncthreads();
    }
    // exactly one j has s_off[j] < E1 <= s_off[j+1]
    for (int j = threadIdx.x; j < limit; j += BLOCK)
      if ((u64)s_off[j] < E1 && (u64)s_off[j + 1] >= E1) s_nseg = j + 1;
    __syncthreads();
    const int nseg = s_nseg;
    const u32 next_off = s_off[nseg];   // start of the first segment not touched by this tile
    const long long seg_next = seg + (((u64)next_off == E1) ? nseg : nseg - 1);
    if (E1 < e_end) prefetch(seg_next);  // lands while the pipeline works on older tiles

    u32 r32[EPT];
    int sj[EPT];
    actC = 0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const u64 r = E0 + (u64)(k * BLOCK + threadIdx.x);
      const bool act = r < E1;
      if (act) actC |= 1u << k;
      r32[k] = act ? (u32)r : (u32)E0;
      sj[k] = 0;
    }
    if (nseg > 1) {
      int top = 1;
      while (top * 2 < nseg) top *= 2;      // largest power of two < nseg (block-uniform)
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
  };

  // Software pipeline over the workgroup's tiles, three tiles deep, one iteration per tile:
  //   S1(it+1) search            (LDS)
  //   S2(it)   col_indices loads (HBM stream)      -> dstB, in flight across the iteration
  //   S3(it-1) snapshot words    (L2-resident)     -> wordA, in flight across the iteration
  //   S4(it-2) candidates -> LDS staging
  // so the two dependent global round trips of a tile overlap the LDS work of the next tiles
  // instead of being paid back to back.
  const int ntiles = (int)((e_end - e_begin + TILE - 1) / TILE);
  int dstA[EPT], dstB[EPT];
  u32 wordA[EPT];
  u32 actA = 0, actB = 0;
#pragma unroll
  for (int k = 0; k < EPT; ++k) { dstA[k] = 0; dstB[k] = 0; wordA[k] = 0xFFFFFFFFu; }
  if (DIAG) dt = (long long)__builtin_readcyclecounter();
  prepare_tile(e_begin);
  MGX_STAMP(0)
  for (int it = 0; it < ntiles + 2; ++it) {
    // ---- S4: tile it-2: snapshot words have landed -> candidates -> LDS staging ----------------
    if (it >= 2) {
      u32 candmask = 0;
#pragma unroll
      for (int k = 0; k < EPT; ++k)
        if (((actA >> k) & 1u) && !(wordA[k] & (1u << (dstA[k] & 31)))) candmask |= 1u << k;
      if (DIAG && (a.flags & 1) && level == (a.flags >> 8)) candmask = 0;
      const int mine = (int)__popc(candmask);
      const int inc = wave_inclusive_sum(mine);
      const int ncand = __shfl(inc, WAVE - 1, WAVE);
      if (ncand) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&s_count, ncand);
        base = __shfl(base, 0, WAVE);
        int pos = base + inc - mine;
#pragma unroll
        for (int k = 0; k < EPT; ++k)
          if (candmask & (1u << k)) st_v[pos++] = (u32)dstA[k];
      }
    }
    MGX_STAMP(1)                         // wait for snapshot words + candidate staging
    // ---- S3: tile it-1: neighbour ids have landed -> issue the snapshot-word gathers -------------
    if (it >= 1 && it - 1 < ntiles) {
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        dstA[k] = dstB[k];
        wordA[k] = a.snapshot[(u32)dstA[k] >> 5];
      }
      actA = actB;
    } else {
      actA = 0;
    }
    MGX_STAMP(2)                         // wait for col_indices
    // ---- S2: tile it: issue the col_indices reads ------------------------------------------------
    if (it < ntiles) {
#pragma unroll
      for (int k = 0; k < EPT; ++k) dstB[k] = a.col_indices[eidxC[k]];
      actB = actC;
    }
    // ---- S1: tile it+1: staging + search ----------------------------------------------------------
    if (it + 1 < ntiles) prepare_tile(e_begin + (u64)(it + 1) * TILE);
    MGX_STAMP(3)                         // staging + search of the next tile
    __syncthreads();
    MGX_STAMP(5)                         // tile-end barrier (waiting for the slowest wave)
    const int cnt = s_count;
    if (cnt >= BFS_FLUSH_AT) flush(cnt);
    MGX_STAMP(6)                         // flush: batched claims + frontier append
  }
  {
    const int cnt = s_count;   // stable: last loop iteration ended with a barrier
    if (cnt > 0) flush(cnt);
  }
  if (threadIdx.x == 0 && s_wins) atomicAdd(&c->reached, (u64)s_wins);
  if (threadIdx.x == 0 && s_claims) {
    atomicAdd(&c->claims, (u64)s_claims);
    if (level < 64) atomicAdd(&c->claims_level[level], (u64)s_claims);
  }
  if (DIAG && threadIdx.x == 0)
    for (int i = 0; i < 8; ++i)
      if (dg[i]) atomicAdd(&c->diag[i], (u64)dg[i]);
#undef MGX_STAMP
}

struct bfs_fused_state_t {
  mem_t<u32> visited;
  mem_t<u32> snapshot;
  mem_t<u32> frontier_bits;
  mem_t<u32> fr_row[2];
  mem_t<u32> fr_off[2];
  mem_t<bfs_ctrl_t> ctrl;
  bfs_ctrl_t* host_ctrl = nullptr;   // pinned copy for stats
  int n = 0;
  int levels_per_sync = 8;
  int grid = 0;
  int ept = 4;                       // edge ranks per lane per tile (4 or 8)
  int occ = 5;                       // workgroups per CU the kernel is register-budgeted for
  bool diag = false;                 // MGX_BFS_DIAG=1: stage-stamped diagnostic kernel (EPT 4)
  int hot_min_tiles = 4;
  int hot = -1;                      // LDS-resident hot bitmap kernel: -1 auto (on when a hub-first layout is attached), 0/1 forced
  // timing of the level kernels of the last run (HIP events around each batch of launches)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double level_kernel_ms = 0.0;
  long long level_kernel_launches = 0;
  float batch_ms[256];               // duration of each launch batch of the last run (per level when levels_per_sync == 1)
  int batches = 0;
  // per-launch timing of the wave-private streaming kernel (the kernel most edges go through)
  static constexpr int EV_POOL = 64;
  hipEvent_t wev[EV_POOL] = {};
  double wave_kernel_ms = 0.0;
  long long wave_kernel_launches = 0;

  bfs_fused_state_t() {}
  bfs_fused_state_t(int num_nodes, standard_context_t& ctx) : n(num_nodes) {
    visited = mem_t<u32>((size_t)(num_nodes + 31) / 32 + 1, ctx);
    {
      size_t words = (size_t)(num_nodes + 31) / 32 + 1;
      if (words < 65536) words = 65536;            // the hot-bitmap kernel copies a fixed 64 KB prefix
      snapshot = mem_t<u32>(words, ctx);
      MGX_HIP(hipMemsetAsync(snapshot.data(), 0, words * sizeof(u32), ctx.stream()));
      frontier_bits = mem_t<u32>(words, ctx);
    }
    for (int i = 0; i < 2; ++i) {
      fr_row[i] = mem_t<u32>((size_t)num_nodes + 1, ctx);
      fr_off[i] = mem_t<u32>((size_t)num_nodes + 1, ctx);
    }
    ctrl = mem_t<bfs_ctrl_t>(1, ctx);
    MGX_HIP(hipHostMalloc((void**)&host_ctrl, sizeof(bfs_ctrl_t), hipHostMallocDefault));
    MGX_HIP(hipEventCreate(&ev0));
    MGX_HIP(hipEventCreate(&ev1));
    for (int i = 0; i < EV_POOL; ++i) MGX_HIP(hipEventCreate(&wev[i]));
    if (const char* e = getenv("MGX_BFS_EPT")) ept = (atoi(e) == 8) ? 8 : 4;
    if (const char* e = getenv("MGX_BFS_DIAG")) diag = atoi(e) != 0;
    if (const char* e = getenv("MGX_BFS_HOT")) hot = atoi(e);
    if (const char* e = getenv("MGX_BFS_HOT_MIN_TILES")) hot_min_tiles = atoi(e);
    occ = (ept == 8) ? 4 : 5;
    if (const char* e = getenv("MGX_BFS_OCC")) occ = atoi(e) > 0 ? atoi(e) : occ;
    grid = ctx.num_cus * occ;
    if (const char* e = getenv("MGX_BFS_LEVELS_PER_SYNC")) levels_per_sync = atoi(e) > 0 ? atoi(e) : 8;
    if (const char* e = getenv("MGX_BFS_GRID")) grid = atoi(e) > 0 ? atoi(e) : grid;
  }
  bfs_fused_state_t(bfs_fused_state_t&& r) noexcept { *this = std::move(r); }
  bfs_fused_state_t& operator=(bfs_fused_state_t&& r) noexcept {
    visited = std::move(r.visited);
    snapshot = std::move(r.snapshot);
    frontier_bits = std::move(r.frontier_bits);
    for (int i = 0; i < 2; ++i) { fr_row[i] = std::move(r.fr_row[i]); fr_off[i] = std::move(r.fr_off[i]); }
    ctrl = std::move(r.ctrl);
    std::swap(host_ctrl, r.host_ctrl);
    std::swap(ev0, r.ev0);
    std::swap(ev1, r.ev1);
    for (int i = 0; i < EV_POOL; ++i) std::swap(wev[i], r.wev[i]);
    n = r.n; levels_per_sync = r.levels_per_sync; grid = r.grid; ept = r.ept; diag = r.diag; occ = r.occ; hot = r.hot; hot_min_tiles = r.hot_min_tiles;
    return *this;
  }
  ~bfs_fused_state_t() {
    if (host_ctrl) (void)hipHostFree(host_ctrl);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    for (int i = 0; i < EV_POOL; ++i) if (wev[i]) (void)hipEventDestroy(wev[i]);
  }
};

}  // namespace mgx
