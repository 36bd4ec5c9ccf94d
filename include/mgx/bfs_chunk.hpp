// mgx/bfs_chunk.hpp -- fused BFS, generation 2: chunk-granular load balancing, wave-private
// streaming, sharded frontier cursors.  Same contract as bfs_fused*.hpp (labels identical, device-
// resident levels, direction-optimising option); selected by bfs_fused_enactor_t.
//
// What the PMC profile of generation 1 said (profiles/): per-EDGE load-balanced search
// costs ~80 VALU instructions per edge and LDS-latency chains the 16-20 resident waves cannot hide
// (68 % of wave cycles parked).  Here the unit of load balancing is a CHUNK = up to 256 consecutive
// edges of ONE row, i.e. one coalesced 1 KB col_indices read by one wave (16 bytes per lane):
//   * the frontier stores (row_start, degree, chunk_offset) per vertex; chunk_offset is the exclusive
//     scan of ceil(degree/256), produced for free by the packed-cursor append as before;
//   * a wave owns a contiguous range of chunk ids.  It keeps a window of 64 frontier rows in
//     REGISTERS (lane t = row j0+t); the row of chunk c is popcount(ballot(chunk_offset <= c)) - 1,
//     one ballot per 256 edges instead of a binary search per edge, and the row's fields come by
//     v_readlane.  No LDS search, no workgroup barrier in the streaming loop;
//   * short rows waste lanes (a 3-edge row still costs a chunk), which is cheap: a chunk is ~20
//     instructions, and skewed graphs keep most edges in long rows;
//   * candidates are staged per wave and claimed in batches; cursor traffic is spread over 8 shards
//     (8 sub-frontiers, consumed as one concatenated chunk space) because one hot 64-bit counter
//     serves only ~83 M returning atomics/s (tools/microbench.hip);
//   * the hot prefix of the level-start visited snapshot sits in LDS (hub-first layout), the rest
//     is probed in L2; the live bitmap is touched only by the batched claims.
#pragma once
#include <cstddef>
#include <type_traits>

#include "bfs_fused_wave.hpp"

namespace mgx {

constexpr int CB_SHARDS = 8;
constexpr int CB_LPE = 4;                        // consecutive edges per lane: one 16-byte col_indices read
constexpr int CB_CHUNK = WAVE * CB_LPE;          // 256 edges of one row = one 1 KB wave read
constexpr int CB_CHUNK_SHIFT = 8;
constexpr int CB_QPT = 2;                        // chunks per wave tile (2 KB of col_indices in flight per wave and stage)
constexpr int CB_FLUSH = 256;                    // staged candidates that trigger a wave flush
constexpr int CB_STAGE = CB_FLUSH + CB_CHUNK;    // 512: the flush check runs after every chunk
constexpr int CB_LDS_PER_WAVE = CB_STAGE * 4;

struct cb_ctrl_t {
  u64 cursor[3][CB_SHARDS];   // level L reads [L%3], appends into [(L+1)%3], clears [(L+2)%3]; (vertices<<38)|chunks
  u64 edges[3];               // sum of degrees appended per level (the TEPS numerator)
  u64 sum_edges, sum_frontier, reached, claims, pull_edges;
  u32 nf_x[CB_SHARDS];        // per-shard frontier sizes of the level about to run
  u32 ch_prefix[CB_SHARDS + 1];   // exclusive prefix of per-shard chunk counts
  int done, levels, pull, push_levels;
  u64 trace[BFS_MAX_TRACE];   // (nf << 38) | edges per level; kept last
};

struct cb_args_t {
  const u32* row_offsets;
  const int* col_indices;
  const u32* in_offsets;
  const int* in_indices;
  int* labels;
  u32* visited;
  u32* snapshot;
  u32* frontier_bits;
  u32* fr_row[2];            // CB_SHARDS lists of `cap` entries each
  u32* fr_deg[2];
  u32* fr_cho[2];
  cb_ctrl_t* ctrl;
  const int* old_of_new;
  const int* new_of_old;
  u32 cap;
  u32 m;                     // entries of col_indices (the 16-byte reads are clamped to m-4)
  int n;
  int hot_min_tiles;
  int mode;
  float alpha;
};

__global__ void k_cb_init(cb_args_t a, int src) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  cb_ctrl_t* c = a.ctrl;
  a.labels[src] = 0;
  if (a.new_of_old) src = a.new_of_old[src];
  a.visited[src >> 5] = 1u << (src & 31);
  const u32 ro = a.row_offsets[src];
  const u32 deg = a.row_offsets[src + 1] - ro;
  for (int l = 0; l < 3; ++l) {
    for (int x = 0; x < CB_SHARDS; ++x) c->cursor[l][x] = 0;
    c->edges[l] = 0;
  }
  if (deg) {
    a.fr_row[0][0] = ro;
    a.fr_deg[0][0] = deg;
    a.fr_cho[0][0] = 0;
    c->cursor[0][0] = (1ull << BFS_VSHIFT) | (u64)((deg + CB_CHUNK - 1) >> CB_CHUNK_SHIFT);
    c->edges[0] = deg;
  }
  c->sum_edges = c->sum_frontier = c->claims = c->pull_edges = 0;
  c->reached = 1;
  c->done = c->levels = c->pull = c->push_levels = 0;
}

// per-level bookkeeping + direction decision + snapshot (+ frontier bitmap for direction-optimising runs)
__global__ __launch_bounds__(BLOCK) void k_cb_begin(cb_args_t a, int level, long long nwords) {
  cb_ctrl_t* const c = a.ctrl;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    u64 nf = 0;
    u32 run = 0;
    for (int x = 0; x < CB_SHARDS; ++x) {
      const u64 cur = c->cursor[level % 3][x];
      c->nf_x[x] = (u32)(cur >> BFS_VSHIFT);
      c->ch_prefix[x] = run;
      run += (u32)(cur & BFS_EMASK);
      nf += cur >> BFS_VSHIFT;
      c->cursor[(level + 2) % 3][x] = 0;
    }
    c->ch_prefix[CB_SHARDS] = run;
    const u64 E = c->edges[level % 3];
    c->edges[(level + 2) % 3] = 0;
    if (nf == 0) {
      if (!c->done) { c->done = 1; c->levels = level; }
    } else {
      if (level < BFS_MAX_TRACE) c->trace[level] = (nf << BFS_VSHIFT) | E;
      c->sum_edges += E;
      c->sum_frontier += nf;
      if (a.mode == 1 && !c->pull) {
        const float unvisited = (float)((long long)a.n - (long long)c->reached);
        if (unvisited < (float)nf * a.alpha) c->pull = 1;        // bfs_enactor.hxx:68, sticky (:74-112)
      }
      if (!c->pull) c->push_levels += 1;
    }
  }
  const bool want_frontier = a.mode == 1;
  for (long long w = (long long)blockIdx.x * BLOCK + threadIdx.x; w < nwords; w += (long long)gridDim.x * BLOCK) {
    const u32 now = a.visited[w];
    if (want_frontier) a.frontier_bits[w] = now & ~a.snapshot[w];
    a.snapshot[w] = now;
  }
}

// Shared by the push and pull kernels: claim (optional) + append the `cnt` vertex ids staged in
// st[0..cnt) of the calling wave.  BW: scan + cursor atomic over the whole workgroup (all waves call
// it together), else wave-private.  CLAIM: ids are candidates to be claimed in the live bitmap
// (push); without it they are already exclusive discoveries (pull).
template <int NW, int PER, bool BW, bool CLAIM>
__device__ __forceinline__ void cb_flush(const cb_args_t& a, int level, int shard, const u32* st, int cnt, u64* s_scan,
                                         u64* s_base, int& wins, int& claims, u64& edges) {
  constexpr u64 CNT1 = 1ull << 40;
  constexpr u64 LOWMASK = CNT1 - 1ull;
  const int lane = lane_id();
  cb_ctrl_t* const c = a.ctrl;
  u32 v[PER], old[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const int i = lane * PER + q;
    v[q] = (i < cnt) ? st[i] : 0u;
    old[q] = 0;
    if (CLAIM) old[q] = a.visited[v[q] >> 5];                  // unconditional: countable loads
  }
  u32 winmask = 0;
  if (CLAIM) {
    u32 livemask = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q)
      if (lane * PER + q < cnt && !(old[q] & (1u << (v[q] & 31)))) livemask |= 1u << q;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(livemask) : : "memory");
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      old[q] = 0xFFFFFFFFu;
      if ((livemask >> q) & 1u) old[q] = atomicOr(a.visited + (v[q] >> 5), 1u << (v[q] & 31));
    }
    claims += wave_sum((int)__popc(livemask));
#pragma unroll
    for (int q = 0; q < PER; ++q)
      if (!(old[q] & (1u << (v[q] & 31)))) winmask |= 1u << q;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(winmask) : : "memory");
  } else {
#pragma unroll
    for (int q = 0; q < PER; ++q)
      if (lane * PER + q < cnt) winmask |= 1u << q;
  }
  u32 ro[PER], ro1[PER];
  int lab_at[PER];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const u32 w = ((winmask >> q) & 1u) ? v[q] : 0u;
    ro[q] = a.row_offsets[w];
    ro1[q] = a.row_offsets[w + 1];
    lab_at[q] = a.old_of_new ? a.old_of_new[w] : (int)w;
  }
#pragma unroll
  for (int q = 0; q < PER; ++q)
    if ((winmask >> q) & 1u) a.labels[lab_at[q]] = level + 1;
  wins += wave_sum((int)__popc(winmask));
  u64 loc[PER];
  u64 sum = 0;
  u64 degsum = 0;
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    const u32 deg = ((winmask >> q) & 1u) ? ro1[q] - ro[q] : 0u;
    loc[q] = sum;
    sum += deg ? (CNT1 | (u64)((deg + CB_CHUNK - 1) >> CB_CHUNK_SHIFT)) : 0ull;
    degsum += deg;
  }
  edges += wave_sum(degsum);
  u64* const cursor = &c->cursor[(level + 1) % 3][shard];
  u64 ex, total, base;
  if (BW) {
    ex = block_exclusive_sum_nw<NW>(sum, s_scan, &total);
    if (threadIdx.x == 0)
      *s_base = (total >> 40) ? atomicAdd(cursor, ((total >> 40) << BFS_VSHIFT) | (total & LOWMASK)) : 0ull;
    __syncthreads();
    base = *s_base;
  } else {
    const u64 inc = wave_inclusive_sum(sum);
    ex = inc - sum;
    total = __shfl(inc, WAVE - 1, WAVE);
    base = 0;
    if (lane == 0 && (total >> 40)) base = atomicAdd(cursor, ((total >> 40) << BFS_VSHIFT) | (total & LOWMASK));
    base = __shfl(base, 0, WAVE);
  }
  const u64 at0 = (u64)shard * a.cap + (base >> BFS_VSHIFT);
  const u32 base_c = (u32)(base & BFS_EMASK);
  u32* __restrict__ o_row = a.fr_row[(level + 1) & 1];
  u32* __restrict__ o_deg = a.fr_deg[(level + 1) & 1];
  u32* __restrict__ o_cho = a.fr_cho[(level + 1) & 1];
#pragma unroll
  for (int q = 0; q < PER; ++q) {
    if (((winmask >> q) & 1u) && ro1[q] != ro[q]) {
      const u64 at = ex + loc[q];
      o_row[at0 + (at >> 40)] = ro[q];
      o_deg[at0 + (at >> 40)] = ro1[q] - ro[q];
      o_cho[at0 + (at >> 40)] = base_c + (u32)(at & LOWMASK);
    }
  }
}

template <int NT, int HOTW>
__global__ __launch_bounds__(NT, 4) void k_cb_push(cb_args_t a, int level) {
  constexpr int NW = NT / WAVE;
  constexpr int PER = 4;                       // staged entries per lane and flush round (256 per round)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32* const hot = (u32*)smem;
  const int wave = threadIdx.x / WAVE;
  const int lane = lane_id();
  u32* const w_st = hot + HOTW + wave * (CB_LDS_PER_WAVE / 4);
  u64* const s_scan = (u64*)(hot + HOTW + NW * (CB_LDS_PER_WAVE / 4));
  u64* const s_base = s_scan + NW + 1;
  u64* const s_edges = s_base + 1;
  int* const s_int = (int*)(s_edges + 1);               // [0] wins [1] claims

  cb_ctrl_t* const c = a.ctrl;
  const u32 Ttot = c->ch_prefix[CB_SHARDS];
  if (Ttot == 0 || c->pull) return;

  const u32* __restrict__ f_row = a.fr_row[level & 1];
  const u32* __restrict__ f_deg = a.fr_deg[level & 1];
  const u32* __restrict__ f_cho = a.fr_cho[level & 1];

  const u32 total_waves = gridDim.x * NW;
  u32 per = (Ttot + total_waves - 1) / total_waves;
  per = (per + CB_QPT - 1) / CB_QPT * CB_QPT;
  const u32 wid = blockIdx.x * NW + wave;
  const u64 cb64 = (u64)wid * per;
  const u32 c_begin = cb64 < (u64)Ttot ? (u32)cb64 : Ttot;
  const u32 c_end = (cb64 + per < (u64)Ttot) ? (u32)(cb64 + per) : Ttot;
  const int shard = (int)(wid & (CB_SHARDS - 1));

  const bool use_hot = per >= (u32)a.hot_min_tiles * (u32)CB_QPT;
  const u32 hot_n = use_hot ? (((u32)a.n < (u32)(HOTW * 32)) ? (u32)a.n : (u32)(HOTW * 32)) : 0u;
  if (use_hot) {
    const uint4* src = (const uint4*)a.snapshot;
    uint4* dstp = (uint4*)hot;
    for (int i = threadIdx.x; i < HOTW / 4; i += NT) dstp[i] = src[i];
  }
  if (threadIdx.x == 0) { s_int[0] = 0; s_int[1] = 0; *s_edges = 0; }
  __syncthreads();

  int count = 0, wins = 0, claims = 0;
  u64 edges = 0;

  // pipeline registers (3 tiles deep, see bfs_fused_hot.hpp): C = addresses, B = col_indices in flight
  // (one 16-byte read per lane and chunk), A = neighbour ids + cold snapshot words in flight
  struct __attribute__((packed, aligned(4))) quad_t { u32 x, y, z, w; };   // dword-aligned 16-byte read
  u32 eidxC[CB_QPT];          // first of the lane's 4 consecutive col_indices entries
  u32 actC[CB_QPT];           // 4-bit validity masks
  u32 dstA[CB_QPT][CB_LPE], dstB[CB_QPT][CB_LPE];
  u32 wordA[CB_QPT][CB_LPE];
  u32 actA[CB_QPT], actB[CB_QPT];
#pragma unroll
  for (int q = 0; q < CB_QPT; ++q) {
    eidxC[q] = 0; actC[q] = 0; actA[q] = 0; actB[q] = 0;
#pragma unroll
    for (int k = 0; k < CB_LPE; ++k) { dstA[q][k] = 0; dstB[q][k] = 0; wordA[q][k] = 0xFFFFFFFFu; }
  }
  bool haveA = false, haveB = false, haveC = false;
  const u32 m_clamp = a.m >= 4u ? a.m - 4u : 0u;

  // one pipeline step: consume A (stage candidates), move B->A, issue C->B; the caller refills C
  auto pipeline_step = [&]() {
    if (haveA) {
#pragma unroll
      for (int q = 0; q < CB_QPT; ++q) {
        u32 candmask = 0;
#pragma unroll
        for (int k = 0; k < CB_LPE; ++k) {
          const u32 d = dstA[q][k];
          const u32 bit = 1u << (d & 31);
          if ((actA[q] >> k) & 1u) {
            if (d < hot_n) {
              if (!(hot[d >> 5] & bit) && !(atomicOr(&hot[d >> 5], bit) & bit)) candmask |= 1u << k;
            } else if (!(wordA[q][k] & bit)) {
              candmask |= 1u << k;
            }
          }
        }
        if (__ballot(candmask != 0)) {
#pragma unroll
          for (int k = 0; k < CB_LPE; ++k) {
            const u64 bal = __ballot((candmask >> k) & 1u);
            if ((candmask >> k) & 1u) w_st[count + rank_in_mask(bal)] = dstA[q][k];
            count += __popcll(bal);
          }
          if (count >= CB_FLUSH) {
            for (int h = 0; h < count; h += PER * WAVE)       // rounds of 256: keeps the flush at 4 entries per lane
              cb_flush<NW, PER, false, true>(a, level, shard, w_st + h, (count - h < PER * WAVE) ? count - h : PER * WAVE,
                                             s_scan, s_base, wins, claims, edges);
            count = 0;
          }
        }
      }
    }
    // unconditional, countable loads (validity travels in the act masks)
#pragma unroll
    for (int q = 0; q < CB_QPT; ++q) {
#pragma unroll
      for (int k = 0; k < CB_LPE; ++k) dstA[q][k] = dstB[q][k];
      actA[q] = haveB ? actB[q] : 0u;
    }
#pragma unroll
    for (int q = 0; q < CB_QPT; ++q) {
      const u32 e = haveC ? eidxC[q] : 0u;
      const u32 el = e < m_clamp ? e : m_clamp;          // never read past the array
      const quad_t t = *(const quad_t*)(a.col_indices + el);
      const u32 sh = e - el;                              // 0 except in the last 3 entries of the array
      u32 v0 = t.x, v1 = t.y, v2 = t.z, v3 = t.w;
      if (sh) {                                           // bring the wanted entries to the front
        v0 = sh == 1 ? t.y : (sh == 2 ? t.z : t.w);
        v1 = sh == 1 ? t.z : t.w;                         // entries past the array end are never valid
        v2 = t.w;
      }
      dstB[q][0] = v0; dstB[q][1] = v1; dstB[q][2] = v2; dstB[q][3] = v3;
      actB[q] = haveC ? actC[q] : 0u;
    }
#pragma unroll
    for (int q = 0; q < CB_QPT; ++q)
#pragma unroll
      for (int k = 0; k < CB_LPE; ++k) {
        const u32 d = ((actA[q] >> k) & 1u) ? dstA[q][k] : 0u;      // inactive entries may hold anything
        dstA[q][k] = d;
        wordA[q][k] = a.snapshot[(d >= hot_n) ? (d >> 5) : 0u];
      }
    haveA = haveB;
    haveB = haveC;
    haveC = false;
  };

  // ---- tile producer: walks the shard lists the wave's chunk range intersects; inside a list keeps a
  //      window of 64 rows in registers (lane t = row j0 + t) and the next window prefetched ------------
  int x = -1;                       // current shard list (-1: none yet)
  u32 cc = 0, lc_end = 0;           // next chunk id / end, local to list x
  u32 nfx = 0, Tx = 0;
  const u32* __restrict__ l_row = f_row;
  const u32* __restrict__ l_deg = f_deg;
  const u32* __restrict__ l_cho = f_cho;
  long long j0 = 0;
  u32 mycho = 0, myrow = 0, mydeg = 0, wend = 0;
  auto load_window = [&](long long j, u32& o_cho, u32& o_row, u32& o_deg, u32& o_end) {
    const long long r = j + lane;
    const bool ok = r < (long long)nfx;
    const long long rc = ok ? r : (long long)nfx - 1;
    const u32 t_cho = l_cho[rc], t_row = l_row[rc], t_deg = l_deg[rc];
    const long long e = j + WAVE;
    const u32 t_end = l_cho[e < (long long)nfx ? e : (long long)nfx - 1];
    o_cho = ok ? t_cho : Tx;               // sentinel: beyond every valid chunk id
    o_row = t_row;
    o_deg = ok ? t_deg : 0u;
    o_end = (e < (long long)nfx) ? t_end : Tx;
  };
  auto next_tile = [&]() -> bool {
    for (;;) {
      if (cc >= lc_end) {                  // list exhausted (or none yet): find the next intersecting one
        bool found = false;
        while (++x < CB_SHARDS) {
          const u32 P0 = c->ch_prefix[x], P1 = c->ch_prefix[x + 1];
          const u32 lo = c_begin > P0 ? c_begin : P0;
          const u32 hi = c_end < P1 ? c_end : P1;
          if (lo >= hi) continue;
          cc = lo - P0;
          lc_end = hi - P0;
          nfx = c->nf_x[x];
          Tx = P1 - P0;
          l_row = f_row + (size_t)x * a.cap;
          l_deg = f_deg + (size_t)x * a.cap;
          l_cho = f_cho + (size_t)x * a.cap;
          j0 = wave_upper_bound(l_cho, (long long)nfx, cc) - 1;
          load_window(j0, mycho, myrow, mydeg, wend);
          found = true;
          break;
        }
        if (!found) { x = CB_SHARDS; cc = lc_end = 0; return false; }
      }
      if (cc >= wend) {                    // all chunks of this window done: slide (one round trip per 64 rows)
        j0 += WAVE;
        load_window(j0, mycho, myrow, mydeg, wend);
        continue;
      }
      const u32 lim = lc_end < wend ? lc_end : wend;
      const u32 nq = (lim - cc < (u32)CB_QPT) ? lim - cc : (u32)CB_QPT;
#pragma unroll
      for (int q = 0; q < CB_QPT; ++q) {
        const u32 cq = cc + (u32)q;
        const int jj = __popcll(__ballot(mycho <= cq)) - 1;      // row of chunk cq (wave-uniform)
        const int js = jj < 0 ? 0 : jj;
        const u32 rcho = (u32)__builtin_amdgcn_readlane((int)mycho, js);
        const u32 rrow = (u32)__builtin_amdgcn_readlane((int)myrow, js);
        const u32 rdeg = (u32)__builtin_amdgcn_readlane((int)mydeg, js);
        const u32 off = (cq - rcho) * (u32)CB_CHUNK + (u32)lane * CB_LPE;
        u32 act = 0;
        if ((u32)q < nq && off < rdeg) {
          const u32 left = rdeg - off;                           // >= 1
          act = left >= 4u ? 0xFu : ((1u << left) - 1u);
        }
        actC[q] = act;
        eidxC[q] = act ? rrow + off : 0u;
      }
      cc += nq;
      return true;
    }
  };
  if (c_begin >= c_end) x = CB_SHARDS;     // no work for this wave
  for (;;) {
    haveC = (x < CB_SHARDS) ? next_tile() : false;
    if (!haveA && !haveB && !haveC) break;
    pipeline_step();
  }

  __syncthreads();
  cb_flush<NW, PER, true, true>(a, level, (int)(blockIdx.x & (CB_SHARDS - 1)), w_st, count, s_scan, s_base, wins,
                                claims, edges);          // count < CB_FLUSH = 256 = PER*64 here
  if (lane == 0) {
    if (wins) atomicAdd(&s_int[0], wins);
    if (claims) atomicAdd(&s_int[1], claims);
    if (edges) atomicAdd((unsigned long long*)s_edges, (unsigned long long)edges);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_int[0]) atomicAdd(&c->reached, (u64)s_int[0]);
    if (s_int[1]) atomicAdd(&c->claims, (u64)s_int[1]);
    if (*s_edges) atomicAdd(&c->edges[(level + 1) % 3], *s_edges);
  }
}

// bottom-up level: as k_bfs_pull_level (bfs_fused_hot.hpp), appending in the chunked format
template <int NT>
__global__ __launch_bounds__(NT) void k_cb_pull(cb_args_t a, int level) {
  constexpr int NW = NT / WAVE;
  constexpr int PER = 2;                       // the workgroup flushes at >= NT staged discoveries: <= 2*NT-1
  __shared__ u32 st_v[PER * NT];
  __shared__ u64 s_scan[NW + 1];
  __shared__ u64 s_base;
  __shared__ unsigned long long s_insp, s_edges;
  __shared__ int s_wins, s_count;
  cb_ctrl_t* const c = a.ctrl;
  if (c->ch_prefix[CB_SHARDS] == 0 || !c->pull) return;
  const int n = a.n;
  long long per_v = ((long long)n + gridDim.x - 1) / gridDim.x;
  per_v = (per_v + NT - 1) / NT * NT;
  const long long v_begin = (long long)blockIdx.x * per_v;
  if (v_begin >= n) return;
  const long long v_end = (v_begin + per_v < n) ? v_begin + per_v : n;
  if (threadIdx.x == 0) { s_wins = 0; s_count = 0; s_insp = 0ull; s_edges = 0ull; }
  __syncthreads();
  const int lane = lane_id();
  const int wave = threadIdx.x / WAVE;
  const int shard = (int)(blockIdx.x & (CB_SHARDS - 1));
  int wins = 0, claims = 0;
  u64 edges = 0;
  unsigned long long insp_total = 0;
  // every wave appends its slice [wave*PER*64, ...) of the workgroup's staging area
  auto flush = [&](int cnt) {
    int mine = cnt - wave * PER * WAVE;
    mine = mine < 0 ? 0 : (mine > PER * WAVE ? PER * WAVE : mine);
    cb_flush<NW, PER, true, false>(a, level, shard, st_v + wave * PER * WAVE, mine, s_scan, &s_base, wins, claims, edges);
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
    insp_total += (unsigned long long)wave_sum(inspected);
    __syncthreads();
    const int cnt = s_count;
    if (cnt >= NT) flush(cnt);
  }
  {
    const int cnt = s_count;
    if (cnt > 0) flush(cnt);
  }
  if (lane == 0) {
    if (wins) atomicAdd(&s_wins, wins);
    if (edges) atomicAdd(&s_edges, (unsigned long long)edges);
    if (insp_total) atomicAdd(&s_insp, insp_total);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_wins) atomicAdd(&c->reached, (u64)s_wins);
    if (s_insp) atomicAdd(&c->pull_edges, (u64)s_insp);
    if (s_edges) atomicAdd(&c->edges[(level + 1) % 3], (u64)s_edges);
  }
}

struct cb_state_t {
  mem_t<u32> visited, snapshot, frontier_bits;
  mem_t<u32> fr_row[2], fr_deg[2], fr_cho[2];
  mem_t<cb_ctrl_t> ctrl;
  cb_ctrl_t* host_ctrl = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  int n = 0;
  u32 cap = 0;
  int levels_per_sync = 8;
  int hot_min_tiles = 4;
  int shape = 0;                     // 0: 512 threads x 48 KB hot x 2 per CU, 1: 1024 threads x 96 KB hot x 1 per CU
  double level_kernel_ms = 0.0;
  long long level_kernel_launches = 0;
  float batch_ms[256];
  int batches = 0;

  cb_state_t() {}
  cb_state_t(const cb_state_t&) = delete;
  cb_state_t& operator=(const cb_state_t&) = delete;
  cb_state_t(int num_nodes, standard_context_t& ctx) : n(num_nodes) {
    size_t words = (size_t)(num_nodes + 31) / 32 + 1;
    if (words < 65536) words = 65536;
    visited = mem_t<u32>(words, ctx);
    snapshot = mem_t<u32>(words, ctx);
    frontier_bits = mem_t<u32>(words, ctx);
    cap = (u32)num_nodes + 64;
    for (int i = 0; i < 2; ++i) {
      fr_row[i] = mem_t<u32>((size_t)cap * CB_SHARDS, ctx);
      fr_deg[i] = mem_t<u32>((size_t)cap * CB_SHARDS, ctx);
      fr_cho[i] = mem_t<u32>((size_t)cap * CB_SHARDS, ctx);
    }
    ctrl = mem_t<cb_ctrl_t>(1, ctx);
    MGX_HIP(hipHostMalloc((void**)&host_ctrl, sizeof(cb_ctrl_t), hipHostMallocDefault));
    MGX_HIP(hipEventCreate(&ev0));
    MGX_HIP(hipEventCreate(&ev1));
    if (const char* e = getenv("MGX_BFS_HOT_MIN_TILES")) hot_min_tiles = atoi(e);
    if (const char* e = getenv("MGX_BFS_LEVELS_PER_SYNC")) levels_per_sync = atoi(e) > 0 ? atoi(e) : 8;
    if (const char* e = getenv("MGX_CB_SHAPE")) shape = atoi(e);
  }
  ~cb_state_t() {
    if (host_ctrl) (void)hipHostFree(host_ctrl);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
  }
};

constexpr size_t cb_lds_bytes(int nt, int hotw) {
  return (size_t)hotw * 4 + (size_t)(nt / 64) * CB_LDS_PER_WAVE + (size_t)(nt / 64 + 1) * 8 + 8 + 8 + 64;
}

inline void cb_run(cb_state_t& st, const int* row_offsets, const int* col_indices, unsigned num_edges, int* labels, int src,
                   standard_context_t& ctx, const bfs_layout_t* layout, int mode, float alpha, const int* in_offsets,
                   const int* in_indices) {
  hipStream_t s = ctx.stream();
  cb_args_t a;
  const bool relabelled = layout && layout->row_offsets;
  a.row_offsets = (const u32*)(relabelled ? layout->row_offsets : row_offsets);
  a.col_indices = relabelled ? layout->col_indices : col_indices;
  a.in_offsets = (const u32*)(relabelled ? layout->row_offsets : (in_offsets ? in_offsets : row_offsets));
  a.in_indices = relabelled ? layout->col_indices : (in_indices ? in_indices : col_indices);
  a.old_of_new = relabelled ? layout->old_of_new : nullptr;
  a.new_of_old = relabelled ? layout->new_of_old : nullptr;
  a.labels = labels;
  a.visited = st.visited.data();
  a.snapshot = st.snapshot.data();
  a.frontier_bits = st.frontier_bits.data();
  for (int i = 0; i < 2; ++i) {
    a.fr_row[i] = st.fr_row[i].data();
    a.fr_deg[i] = st.fr_deg[i].data();
    a.fr_cho[i] = st.fr_cho[i].data();
  }
  a.ctrl = st.ctrl.data();
  a.cap = st.cap;
  a.m = num_edges;
  a.n = st.n;
  a.hot_min_tiles = st.hot_min_tiles;
  a.mode = mode;
  a.alpha = alpha;
  static bool attr_set = false;
  if (!attr_set) {
    MGX_HIP(hipFuncSetAttribute((const void*)k_cb_push<512, 12288>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGX_HIP(hipFuncSetAttribute((const void*)k_cb_push<1024, 24576>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  MGX_HIP(hipMemsetAsync(labels, 0xFF, (size_t)st.n * sizeof(int), s));
  MGX_HIP(hipMemsetAsync(st.visited.data(), 0, st.visited.size() * sizeof(u32), s));
  MGX_HIP(hipMemsetAsync(st.snapshot.data(), 0, st.snapshot.size() * sizeof(u32), s));
  hipLaunchKernelGGL(k_cb_init, dim3(1), dim3(64), 0, s, a, src);
  const long long nwords = ((long long)st.n + 31) / 32;
  int level = 0;
  st.level_kernel_ms = 0.0;
  st.level_kernel_launches = 0;
  st.batches = 0;
  for (;;) {
    MGX_HIP(hipEventRecord(st.ev0, s));
    for (int i = 0; i < st.levels_per_sync; ++i, ++level) {
      hipLaunchKernelGGL(k_cb_begin, dim3(grid_for(nwords, BLOCK, 256)), dim3(BLOCK), 0, s, a, level, nwords);
      if (st.shape == 1)
        hipLaunchKernelGGL((k_cb_push<1024, 24576>), dim3(ctx.num_cus), dim3(1024), cb_lds_bytes(1024, 24576), s, a, level);
      else
        hipLaunchKernelGGL((k_cb_push<512, 12288>), dim3(ctx.num_cus * 2), dim3(512), cb_lds_bytes(512, 12288), s, a, level);
      if (mode == 1) hipLaunchKernelGGL(k_cb_pull<256>, dim3(ctx.num_cus * 8), dim3(256), 0, s, a, level);
    }
    MGX_HIP(hipEventRecord(st.ev1, s));
    MGX_HIP(hipMemcpyAsync(&st.host_ctrl->done, &st.ctrl.data()->done, sizeof(int), hipMemcpyDeviceToHost, s));
    MGX_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    MGX_HIP(hipEventElapsedTime(&ms, st.ev0, st.ev1));
    st.level_kernel_ms += ms;
    if (st.batches < 256) st.batch_ms[st.batches++] = ms;
    st.level_kernel_launches += st.levels_per_sync;
    if (st.host_ctrl->done) break;
  }
  MGX_HIP(hipMemcpyAsync(st.host_ctrl, st.ctrl.data(), offsetof(cb_ctrl_t, trace), hipMemcpyDeviceToHost, s));
  MGX_HIP(hipStreamSynchronize(s));
  const int lv = st.host_ctrl->levels < BFS_MAX_TRACE ? st.host_ctrl->levels : BFS_MAX_TRACE;
  if (lv > 0) {
    MGX_HIP(hipMemcpyAsync(st.host_ctrl->trace, st.ctrl.data()->trace, (size_t)lv * sizeof(u64), hipMemcpyDeviceToHost, s));
    MGX_HIP(hipStreamSynchronize(s));
  }
}

}  // namespace mgx
