// mgx/bfs_fused_wave.hpp -- wave-private streaming variant of the fused push level.
//
// Measured on the workgroup-synchronous kernels (rocprofv3 PMC, profiles/): the waves of the big
// RMAT level spend 68 % of their cycles parked (SQ_WAIT_ANY) -- 3-4 workgroup barriers per tile, each
// waiting for the slowest of the waves, around LDS-latency chains -- and ~80 VALU instructions per
// edge.  In such a level almost every edge leads to an already visited vertex, so there is nothing
// to batch per workgroup.  Here every WAVE owns a contiguous slice of the level's edge ranks and
// streams it on its own: its (offset,row) slice, its search and its candidate staging live in a
// wave-private LDS region, cross-lane steps are ballots/shuffles, and there is no barrier between the
// initial copy of the hot bitmap and the final cooperative flush.  Ranks are 32-bit throughout.
//
// Candidates (snapshot misses, after the exact intra-workgroup dedup in the LDS hot bitmap) are
// staged per wave; a wave that has collected WFLUSH of them claims and appends them itself (4 round
// trips, the other waves of the CU keep streaming); what is left at the end of the slice is flushed
// by the whole workgroup with ONE packed cursor atomic, so a level costs a few thousand cursor
// atomics instead of one per wave per tile.
// Used for levels whose average frontier degree is small (k_bfs_level_begin decides); discovery-heavy
// levels (a few hub rows) stay on the workgroup-synchronous kernel with its large batched flushes.
#pragma once
#include "bfs_fused_hot.hpp"

namespace mgx {

constexpr int BFS_WAVE_EPT = 4;                       // ranks per lane per wave tile
constexpr int BFS_WAVE_TILE = WAVE * BFS_WAVE_EPT;    // 256 ranks
constexpr int BFS_WAVE_FLUSH = 256;                   // staged candidates that trigger a wave flush
constexpr int BFS_WAVE_STAGE = BFS_WAVE_FLUSH + BFS_WAVE_TILE;   // 512 slots per wave
constexpr int BFS_WAVE_LDS_PER_WAVE = (68 + 64 + BFS_WAVE_STAGE) * 4;

constexpr size_t bfs_wave_lds_bytes(int nt, int hotw) {
  return (size_t)hotw * 4 + (size_t)(nt / 64) * BFS_WAVE_LDS_PER_WAVE + (size_t)(nt / 64 + 1) * 8 + 64;
}

template <int NT, int HOTW>
__global__ __launch_bounds__(NT) void k_bfs_push_level_wave(bfs_fused_args_t a, int level) {
  constexpr int NW = NT / WAVE;
  constexpr int EPT = BFS_WAVE_EPT;
  constexpr int PER = BFS_WAVE_STAGE / WAVE;          // 8 staged entries per lane in a flush
  constexpr u64 CNT1 = 1ull << 40;
  constexpr u64 DEGMASK = CNT1 - 1ull;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32* const hot = (u32*)smem;
  const int wave = threadIdx.x / WAVE;
  const int lane = lane_id();
  u32* const w_off = hot + HOTW + wave * (BFS_WAVE_LDS_PER_WAVE / 4);   // 65 used (+3 pad)
  u32* const w_row = w_off + 68;                                         // 64
  u32* const w_st = w_row + 64;                                          // BFS_WAVE_STAGE
  u64* const s_scan = (u64*)(hot + HOTW + NW * (BFS_WAVE_LDS_PER_WAVE / 4));   // NW + 1
  u64* const s_base = s_scan + NW + 1;
  int* const s_int = (int*)(s_base + 1);               // [0] wins [1] claims

  bfs_ctrl_t* const c = a.ctrl;
  const u64 cur = c->cursor[level % 3];
  const long long nf = (long long)(cur >> BFS_VSHIFT);
  const u32 E = (u32)(cur & BFS_EMASK);
  if (nf == 0 || c->pull || c->kind != 1) return;      // k_bfs_level_begin: bookkeeping, direction, kernel

  const u32* __restrict__ fr_row = a.fr_row[level & 1];
  const u32* __restrict__ fr_off = a.fr_off[level & 1];
  u32* __restrict__ out_row = a.fr_row[(level + 1) & 1];
  u32* __restrict__ out_off = a.fr_off[(level + 1) & 1];
  u64* const out_cursor = &c->cursor[(level + 1) % 3];
  const int* __restrict__ old_of_new = a.old_of_new;
  const int new_label = level + 1;

  // slice of this wave
  const u32 total_waves = gridDim.x * NW;
  u32 per = (E + total_waves - 1) / total_waves;
  per = (per + BFS_WAVE_TILE - 1) / BFS_WAVE_TILE * BFS_WAVE_TILE;
  const u64 rb = (u64)(blockIdx.x * NW + wave) * per;
  const bool has_work = rb < (u64)E;
  const u32 r_begin = has_work ? (u32)rb : E;
  const u32 r_end = (rb + per < (u64)E) ? (u32)(rb + per) : E;

  // the hot prefix of the snapshot (grid-uniform decision: worth it only for big levels)
  const bool use_hot = per >= (u32)a.hot_min_tiles * (u32)BFS_WAVE_TILE;
  const u32 hot_n = use_hot ? (((u32)a.n < (u32)(HOTW * 32)) ? (u32)a.n : (u32)(HOTW * 32)) : 0u;
  if (use_hot) {
    const uint4* src = (const uint4*)a.snapshot;
    uint4* dstp = (uint4*)hot;
    for (int i = threadIdx.x; i < HOTW / 4; i += NT) dstp[i] = src[i];
  }
  if (threadIdx.x == 0) { s_int[0] = 0; s_int[1] = 0; }
  __syncthreads();

  int count = 0;       // staged candidates of this wave (wave-uniform)
  int wins = 0, claims = 0;

  // ---- claim + append `cnt` staged candidates.  BLOCKWIDE: scan and cursor atomic over the whole
  //      workgroup (every wave calls it once, at the end); otherwise wave-private. -----------------------
  auto flush = [&](int cnt, auto blockwide) {
    constexpr bool BW = decltype(blockwide)::value;
    u32 v[PER], old[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = lane * PER + q;
      v[q] = (i < cnt) ? w_st[i] : 0u;
      old[q] = a.visited[v[q] >> 5];                   // unconditional: countable loads
    }
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
    u32 winmask = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q)
      if (!(old[q] & (1u << (v[q] & 31)))) winmask |= 1u << q;
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(winmask) : : "memory");
    if (!a.append) { wins += wave_sum((int)__popc(winmask)); return; }
    u32 ro[PER], ro1[PER];
    int lab_at[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const u32 w = ((winmask >> q) & 1u) ? v[q] : 0u;
      ro[q] = a.row_offsets[w];
      ro1[q] = a.row_offsets[w + 1];
      lab_at[q] = old_of_new ? old_of_new[w] : (int)w;
    }
#pragma unroll
    for (int q = 0; q < PER; ++q)
      if ((winmask >> q) & 1u) a.labels[lab_at[q]] = new_label;
    wins += wave_sum((int)__popc(winmask));
    u64 loc[PER];
    u64 sum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const u32 deg = ((winmask >> q) & 1u) ? ro1[q] - ro[q] : 0u;
      loc[q] = sum;
      sum += deg ? (CNT1 | (u64)deg) : 0ull;
    }
    u64 ex, total, base;
    if (BW) {
      ex = block_exclusive_sum_nw<NW>(sum, s_scan, &total);
      if (threadIdx.x == 0)
        *s_base = (total >> 40) ? atomicAdd(out_cursor, ((total >> 40) << BFS_VSHIFT) | (total & DEGMASK)) : 0ull;
      __syncthreads();
      base = *s_base;
    } else {
      const u64 inc = wave_inclusive_sum(sum);
      ex = inc - sum;
      total = __shfl(inc, WAVE - 1, WAVE);
      base = 0;
      if (lane == 0 && (total >> 40))
        base = atomicAdd(out_cursor, ((total >> 40) << BFS_VSHIFT) | (total & DEGMASK));
      base = __shfl(base, 0, WAVE);
    }
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
  };

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
      const u32 nxt = w_off[lane + 1];                     // same-wave LDS traffic is ordered
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
      // ---- S4: tile it-2: dedup in the LDS hot bitmap / test the cold snapshot word, stage ------------
      if (haveA) {
        u32 candmask = 0;
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
          const u32 d = (u32)dstA[k];
          const u32 bit = 1u << (d & 31);
          if ((actA >> k) & 1u) {
            if (d < hot_n) {
              if (!(hot[d >> 5] & bit) && !(atomicOr(&hot[d >> 5], bit) & bit)) candmask |= 1u << k;
            } else if (!(wordA[k] & bit)) {
              candmask |= 1u << k;
            }
          }
        }
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
          const u64 bal = __ballot((candmask >> k) & 1u);
          if ((candmask >> k) & 1u) w_st[count + rank_in_mask(bal)] = (u32)dstA[k];
          count += __popcll(bal);
        }
        if (count >= BFS_WAVE_FLUSH) {
          flush(count, std::false_type());
          count = 0;
        }
      }
      // ---- S3a / S2 / S3b: unconditional, countable loads (see bfs_fused_hot.hpp) -------------------------
#pragma unroll
      for (int k = 0; k < EPT; ++k) dstA[k] = dstB[k];
      actA = haveB ? actB : 0u;
#pragma unroll
      for (int k = 0; k < EPT; ++k) dstB[k] = a.col_indices[haveC ? eidxC[k] : 0u];
      actB = haveC ? actC : 0u;
#pragma unroll
      for (int k = 0; k < EPT; ++k)
        wordA[k] = a.snapshot[((u32)dstA[k] >= hot_n) ? ((u32)dstA[k] >> 5) : 0u];
      haveA = haveB;
      haveB = haveC;
      // ---- S1: tile it+1 -------------------------------------------------------------------------------------
      haveC = e_next < r_end;
      if (haveC) prepare_tile();
    }
  }

  // ---- the leftovers of all waves: one cooperative flush, one cursor atomic per workgroup -----------------
  __syncthreads();
  flush(count, std::true_type());
  if (lane == 0) {
    if (wins) atomicAdd(&s_int[0], wins);
    if (claims) atomicAdd(&s_int[1], claims);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    if (s_int[0]) atomicAdd(&c->reached, (u64)s_int[0]);
    if (s_int[1]) {
      atomicAdd(&c->claims, (u64)s_int[1]);
      if (level < 64) atomicAdd(&c->claims_level[level], (u64)s_int[1]);
    }
  }
}

// layout (optional): a hub-first relabelled copy of the CSR plus the two id maps; labels stay in the
// original id space either way.
struct bfs_layout_t {
  const int* row_offsets = nullptr;
  const int* col_indices = nullptr;
  const int* new_of_old = nullptr;
  const int* old_of_new = nullptr;
};

// mode/alpha: MGX_BFS_PUSH (0) or MGX_BFS_DIRECTION_OPT (1) with the reference's switch rule
// num_unvisited < frontier_length * alpha (bfs_enactor.hxx:68).  in_offsets/in_indices: in-edges for the
// bottom-up levels (pass the CSR for symmetric graphs, the reference's behaviour -- SURVEY F8).
inline void bfs_fused_run(bfs_fused_state_t& st, const int* row_offsets, const int* col_indices, int* labels,
                          int src, standard_context_t& ctx, const bfs_layout_t* layout = nullptr, int mode = 0,
                          float alpha = 0.f, const int* in_offsets = nullptr, const int* in_indices = nullptr) {
  hipStream_t s = ctx.stream();
  bfs_fused_args_t a;
  const bool relabelled = layout && layout->row_offsets;
  a.row_offsets = (const u32*)(relabelled ? layout->row_offsets : row_offsets);
  a.col_indices = relabelled ? layout->col_indices : col_indices;
  a.old_of_new = relabelled ? layout->old_of_new : nullptr;
  a.new_of_old = relabelled ? layout->new_of_old : nullptr;
  const bool hot = (st.hot < 0) ? relabelled : (st.hot != 0);
  // hot-kernel shapes: 0 = 512 threads x 64 KB bitmap x 2 per CU; 1 = 256 threads x 32 KB x 4 per CU;
  // 2 = 256 threads, 8 ranks per lane, 32 KB x 4 per CU
  static int shape = getenv("MGX_BFS_HOT_SHAPE") ? atoi(getenv("MGX_BFS_HOT_SHAPE")) : 1;
  const int grid = hot ? ctx.num_cus * (shape == 0 ? 2 : 4) : st.grid;
  const size_t hot_lds = shape == 0 ? bfs_hot_lds_bytes(512, 4, 16384)
                                    : (shape == 1 ? bfs_hot_lds_bytes(256, 4, 8192) : bfs_hot_lds_bytes(256, 8, 8192));
  if (hot) {
    static bool attr_set = false;
    if (!attr_set) {
#define MGX_SET_LDS(K_) MGX_HIP(hipFuncSetAttribute((const void*)K_, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
      MGX_SET_LDS((k_bfs_push_level_hot<512, 4, 16384, false>));
      MGX_SET_LDS((k_bfs_push_level_hot<512, 4, 16384, true>));
      MGX_SET_LDS((k_bfs_push_level_hot<256, 4, 8192, false>));
      MGX_SET_LDS((k_bfs_push_level_hot<256, 8, 8192, false>));
      MGX_SET_LDS((k_bfs_push_level_wave<512, 12288>));
      MGX_SET_LDS((k_bfs_push_level_wave<1024, 24576>));
#undef MGX_SET_LDS
      attr_set = true;
    }
  }
  a.labels = labels;
  a.visited = st.visited.data();
  a.snapshot = st.snapshot.data();
  a.frontier_bits = st.frontier_bits.data();
  a.mode = mode;
  a.alpha = alpha;
  a.in_offsets = (const u32*)(relabelled ? layout->row_offsets : (in_offsets ? in_offsets : row_offsets));
  a.in_indices = relabelled ? layout->col_indices : (in_indices ? in_indices : col_indices);
  for (int i = 0; i < 2; ++i) { a.fr_row[i] = st.fr_row[i].data(); a.fr_off[i] = st.fr_off[i].data(); }
  a.ctrl = st.ctrl.data();
  a.n = st.n;
  a.hot_min_tiles = st.hot_min_tiles;
  a.append = 1;
  static int wave_shape = getenv("MGX_BFS_WAVE") ? atoi(getenv("MGX_BFS_WAVE")) : 1;   // 0 off, 1: 512 thr x 48 KB, 2: 1024 thr x 96 KB
  a.wave_kernel = (hot && wave_shape != 0) ? 1 : 0;
  a.wave_max_avg_degree = getenv("MGX_BFS_WAVE_MAX_DEG") ? atoi(getenv("MGX_BFS_WAVE_MAX_DEG")) : 512;
  a.flags = 0;
  if (const char* e = getenv("MGX_BFS_FLAGS")) a.flags = atoi(e);
  MGX_HIP(hipMemsetAsync(labels, 0xFF, (size_t)st.n * sizeof(int), s));
  MGX_HIP(hipMemsetAsync(st.visited.data(), 0, st.visited.size() * sizeof(u32), s));
  // the snapshot must start empty: level 0's frontier bitmap is (visited & ~snapshot)
  MGX_HIP(hipMemsetAsync(st.snapshot.data(), 0, st.snapshot.size() * sizeof(u32), s));
  hipLaunchKernelGGL(k_bfs_fused_init, dim3(1), dim3(64), 0, s, a, src);
  int level = 0;
  st.level_kernel_ms = 0.0;
  st.level_kernel_launches = 0;
  st.wave_kernel_ms = 0.0;
  st.wave_kernel_launches = 0;
  st.batches = 0;
  for (;;) {
    MGX_HIP(hipEventRecord(st.ev0, s));
    for (int i = 0; i < st.levels_per_sync; ++i, ++level) {
      // bookkeeping + direction decision + level-start snapshot of the visited bitmap (n/8 bytes)
      {
        const long long nwords = ((long long)st.n + 31) / 32;
        hipLaunchKernelGGL(k_bfs_level_begin, dim3(grid_for(nwords, BLOCK, 256)), dim3(BLOCK), 0, s, a, level, nwords);
      }
#define MGX_LAUNCH_LEVEL(E_, O_, D_) \
  hipLaunchKernelGGL((k_bfs_push_level<E_, O_, D_>), dim3(st.grid), dim3(BLOCK), 0, s, a, level)
      if (hot && st.diag)
        hipLaunchKernelGGL((k_bfs_push_level_hot<512, 4, 16384, true>), dim3(ctx.num_cus * 2), dim3(512),
                           bfs_hot_lds_bytes(512, 4, 16384), s, a, level);
      else if (hot && shape == 0)
        hipLaunchKernelGGL((k_bfs_push_level_hot<512, 4, 16384, false>), dim3(grid), dim3(512), hot_lds, s, a, level);
      else if (hot && shape == 1)
        hipLaunchKernelGGL((k_bfs_push_level_hot<256, 4, 8192, false>), dim3(grid), dim3(256), hot_lds, s, a, level);
      else if (hot)
        hipLaunchKernelGGL((k_bfs_push_level_hot<256, 8, 8192, false>), dim3(grid), dim3(256), hot_lds, s, a, level);
      else if (st.diag) MGX_LAUNCH_LEVEL(4, 5, true);
      else if (st.ept == 8 && st.occ <= 3) MGX_LAUNCH_LEVEL(8, 3, false);
      else if (st.ept == 8) MGX_LAUNCH_LEVEL(8, 4, false);
      else if (st.occ >= 6) MGX_LAUNCH_LEVEL(4, 6, false);
      else if (st.occ == 5) MGX_LAUNCH_LEVEL(4, 5, false);
      else MGX_LAUNCH_LEVEL(4, 4, false);
#undef MGX_LAUNCH_LEVEL
      const bool time_wave = a.wave_kernel && 2 * i + 1 < bfs_fused_state_t::EV_POOL;
      if (time_wave) MGX_HIP(hipEventRecord(st.wev[2 * i], s));
      if (a.wave_kernel && wave_shape == 2)
        hipLaunchKernelGGL((k_bfs_push_level_wave<1024, 24576>), dim3(ctx.num_cus), dim3(1024),
                           bfs_wave_lds_bytes(1024, 24576), s, a, level);
      else if (a.wave_kernel)
        hipLaunchKernelGGL((k_bfs_push_level_wave<512, 12288>), dim3(ctx.num_cus * 2), dim3(512),
                           bfs_wave_lds_bytes(512, 12288), s, a, level);
      if (time_wave) MGX_HIP(hipEventRecord(st.wev[2 * i + 1], s));
      if (mode == 1)
        hipLaunchKernelGGL(k_bfs_pull_level<256>, dim3(ctx.num_cus * 8), dim3(256), 0, s, a, level);
    }
    MGX_HIP(hipEventRecord(st.ev1, s));
    MGX_HIP(hipMemcpyAsync(&st.host_ctrl->done, &st.ctrl.data()->done, sizeof(int), hipMemcpyDeviceToHost, s));
    MGX_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    MGX_HIP(hipEventElapsedTime(&ms, st.ev0, st.ev1));
    st.level_kernel_ms += ms;
    if (a.wave_kernel)
      for (int i = 0; i < st.levels_per_sync && 2 * i + 1 < bfs_fused_state_t::EV_POOL; ++i) {
        float wms = 0.f;
        MGX_HIP(hipEventElapsedTime(&wms, st.wev[2 * i], st.wev[2 * i + 1]));
        st.wave_kernel_ms += wms;
        st.wave_kernel_launches += 1;
      }
    if (st.batches < 256) st.batch_ms[st.batches++] = ms;
    st.level_kernel_launches += st.levels_per_sync;
    if (st.host_ctrl->done) break;
  }
  // counters first, then only the part of the per-level trace that was written
  MGX_HIP(hipMemcpyAsync(st.host_ctrl, st.ctrl.data(), offsetof(bfs_ctrl_t, trace), hipMemcpyDeviceToHost, s));
  MGX_HIP(hipStreamSynchronize(s));
  const int lv = st.host_ctrl->levels < BFS_MAX_TRACE ? st.host_ctrl->levels : BFS_MAX_TRACE;
  if (lv > 0) {
    MGX_HIP(hipMemcpyAsync(st.host_ctrl->trace, st.ctrl.data()->trace, (size_t)lv * sizeof(u64), hipMemcpyDeviceToHost, s));
    MGX_HIP(hipStreamSynchronize(s));
  }
}

}  // namespace mgx
