// mgx/bfs_dist2.hpp -- partitioned BFS, generation 2: every rank runs the FUSED level kernels
// (bfs_fused_hot.hpp / bfs_fused_wave.hpp) on its rows; ranks exchange dense "newly visited" bitmaps.
//
// Generation 1 (bfs_dist.hpp) expands with the operator-path scan + LBS kernels and ships per-owner id
// lists: ~10 GTEPS per rank on RMAT-22, a ninth of the single-GPU fused path.  Here:
//   * ids are GLOBAL and hub-first (renumbered by descending global degree, as in the single-GPU layout);
//     vertex v is owned by rank v % G (cyclic, so every rank holds its share of the hubs), local row v / G;
//   * every rank keeps the visited bitmap of ALL n vertices (n/8 bytes: 4 MB at RMAT-25).  A level runs the
//     fused push kernels in claims-only mode (args.append = 0): snapshot test in LDS/L2, batched atomicOr
//     claims into the rank's live bitmap -- nothing is appended, no owner is consulted;
//   * new_bits = live & ~snapshot is the rank's discoveries of the level (any owner).  One all-gather of
//     these bitmaps per level replaces the id exchange: volume n/8 bytes per rank and level whatever the
//     frontier size, all 7 xGMI links busy, no counts to exchange first;
//   * merge: OR of the G bitmaps = the level's global discoveries; every rank ORs them into its live
//     bitmap (so all ranks agree before the next level) and turns the bits it OWNS into labels and into
//     its next local frontier (row start, scanned degree; packed-cursor append as everywhere else).
// Labels are the global BFS depths, identical to the single-GPU result.
#pragma once
#include <cstddef>
#include <memory>

#include "bfs_fused_wave.hpp"

namespace mgx {

__global__ __launch_bounds__(BLOCK) void k_d2_newbits(const u32* __restrict__ visited, const u32* __restrict__ snapshot,
                                                      u32* __restrict__ out, long long nwords, bfs_ctrl_t* c) {
  if (blockIdx.x == 0 && threadIdx.x == 0) c->merged_new = 0;     // k_d2_or of this level counts into it
  for (long long w = (long long)blockIdx.x * BLOCK + threadIdx.x; w < nwords; w += (long long)gridDim.x * BLOCK)
    out[w] = visited[w] & ~snapshot[w];
}

// merged[w] = OR over ranks of gathered[r][w]; visited |= merged; ctrl->merged_new += popcount(merged).
// Every rank computes the same count, so the traversal ends on all ranks together without a reduction.
__global__ __launch_bounds__(BLOCK) void k_d2_or(const u32* __restrict__ gathered, int ranks, long long nwords,
                                                 u32* __restrict__ merged, u32* __restrict__ visited, bfs_ctrl_t* c) {
  int found = 0;
  for (long long w = (long long)blockIdx.x * BLOCK + threadIdx.x; w < nwords; w += (long long)gridDim.x * BLOCK) {
    u32 g = 0;
    for (int r = 0; r < ranks; ++r) g |= gathered[(long long)r * nwords + w];
    merged[w] = g;
    if (g) visited[w] |= g;
    found += __popc(g);
  }
  found = wave_sum(found);
  if (lane_id() == 0 && found) atomicAdd(&c->merged_new, (u64)found);
}

// owned vertices whose bit is set in `merged`: label them and append them to the next local frontier
template <int NT>
__global__ __launch_bounds__(NT) void k_d2_build(bfs_fused_args_t a, const u32* __restrict__ merged, int* __restrict__ labels_local,
                                                 int n_local, int ranks, int rank, int level) {
  constexpr int NW = NT / WAVE;
  constexpr int STAGE = 2 * NT;
  constexpr int PER = STAGE / NT;
  constexpr u64 CNT1 = 1ull << 40;
  constexpr u64 DEGMASK = CNT1 - 1ull;
  __shared__ u32 st_v[STAGE];
  __shared__ u64 s_scan[NW + 1];
  __shared__ u64 s_base;
  __shared__ int s_count;
  bfs_ctrl_t* const c = a.ctrl;
  u32* __restrict__ out_row = a.fr_row[(level + 1) & 1];
  u32* __restrict__ out_off = a.fr_off[(level + 1) & 1];
  u64* const out_cursor = &c->cursor[(level + 1) % 3];
  long long per_v = ((long long)n_local + gridDim.x - 1) / gridDim.x;
  per_v = (per_v + NT - 1) / NT * NT;
  const long long i_begin = (long long)blockIdx.x * per_v;
  if (i_begin >= n_local) return;
  const long long i_end = (i_begin + per_v < n_local) ? i_begin + per_v : n_local;
  if (threadIdx.x == 0) s_count = 0;
  __syncthreads();
  const int lane = lane_id();
  auto flush = [&](int cnt) {
    u32 li[PER], ro[PER], ro1[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = threadIdx.x * PER + q;
      li[q] = (i < cnt) ? st_v[i] : 0u;
      ro[q] = a.row_offsets[li[q]];
      ro1[q] = a.row_offsets[li[q] + 1];
    }
    u64 loc[PER];
    u64 sum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = threadIdx.x * PER + q;
      const u32 deg = (i < cnt) ? ro1[q] - ro[q] : 0u;
      loc[q] = sum;
      sum += deg ? (CNT1 | (u64)deg) : 0ull;
    }
    u64 total;
    const u64 ex = block_exclusive_sum_nw<NW>(sum, s_scan, &total);
    if (threadIdx.x == 0)
      s_base = (total >> 40) ? atomicAdd(out_cursor, ((total >> 40) << BFS_VSHIFT) | (total & DEGMASK)) : 0ull;
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
  for (long long base = i_begin; base < i_end; base += NT) {
    const long long i = base + threadIdx.x;
    bool found = false;
    if (i < i_end) {
      const long long v = i * ranks + rank;                 // global id of local row i
      found = (merged[v >> 5] >> (v & 31)) & 1u;
      if (found) labels_local[i] = level + 1;
    }
    const u64 bal = __ballot(found);
    const int nfound = __popcll(bal);
    if (nfound) {
      int at = 0;
      if (lane == 0) at = atomicAdd(&s_count, nfound);
      at = __builtin_amdgcn_readfirstlane(at);
      if (found) st_v[at + rank_in_mask(bal)] = (u32)i;
    }
    __syncthreads();
    const int cnt = s_count;
    if (cnt >= NT) flush(cnt);
  }
  {
    const int cnt = s_count;
    if (cnt > 0) flush(cnt);
  }
}

__global__ void k_d2_init(bfs_fused_args_t a, int* labels_local, int src, int ranks, int rank) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  bfs_ctrl_t* c = a.ctrl;
  a.visited[src >> 5] = 1u << (src & 31);      // every rank knows the source is visited
  c->cursor[0] = c->cursor[1] = c->cursor[2] = 0;
  if (src % ranks == rank) {
    const int i = src / ranks;
    labels_local[i] = 0;
    const u32 ro = a.row_offsets[i];
    const u32 deg = a.row_offsets[i + 1] - ro;
    a.fr_row[0][0] = ro;
    a.fr_off[0][0] = 0;
    c->cursor[0] = deg ? ((1ull << BFS_VSHIFT) | (u64)deg) : 0ull;
  }
  c->sum_edges = c->sum_frontier = c->claims = 0;
  c->reached = 1;
  c->done = c->levels = c->pull = c->push_levels = c->kind = 0;
  c->kind_mask = 0;
  c->pull_edges = 0;
  for (int i = 0; i < 64; ++i) c->claims_level[i] = 0;
}

struct d2_state_t {
  int n_global = 0, n_local = 0, ranks = 1, rank = 0;
  const int* row_offsets = nullptr;   // local rows (n_local + 1)
  const int* col_indices = nullptr;   // global hub-first ids
  std::unique_ptr<bfs_fused_state_t> fs;
  mem_t<int> labels;                  // n_local
  mem_t<u32> merged;                  // n_global bits
  u32* newbits = nullptr;             // caller-owned (n_global bits): what this rank discovered in the level
  long long nwords = 0;
  long long last_edges = 0;           // edges of the frontier the last push expanded

  void init(standard_context_t& ctx, int n_global_, int ranks_, int rank_, const int* ro, const int* ci, u32* newbits_) {
    n_global = n_global_; ranks = ranks_; rank = rank_;
    n_local = (n_global - rank + ranks - 1) / ranks;
    row_offsets = ro; col_indices = ci; newbits = newbits_;
    nwords = ((long long)n_global + 31) / 32;
    fs.reset(new bfs_fused_state_t(n_global, ctx));
    labels = mem_t<int>((size_t)n_local + 1, ctx);
    merged = mem_t<u32>((size_t)nwords + 1, ctx);
  }
  bfs_fused_args_t args() const {
    bfs_fused_args_t a;
    a.row_offsets = (const u32*)row_offsets;
    a.col_indices = col_indices;
    a.labels = nullptr;
    a.visited = fs->visited.data();
    a.snapshot = fs->snapshot.data();
    a.frontier_bits = fs->frontier_bits.data();
    a.in_offsets = nullptr; a.in_indices = nullptr;
    for (int i = 0; i < 2; ++i) { a.fr_row[i] = fs->fr_row[i].data(); a.fr_off[i] = fs->fr_off[i].data(); }
    a.ctrl = fs->ctrl.data();
    a.old_of_new = nullptr; a.new_of_old = nullptr;
    a.n = n_global;
    a.hot_min_tiles = fs->hot_min_tiles;
    a.wave_kernel = 1;
    a.wave_max_avg_degree = 512;
    a.append = 0;
    a.mode = 0; a.alpha = 0.f;
    a.flags = 0;
    return a;
  }
};

// Returns the number of edges of this rank's level-0 frontier (the source's degree on its owner, else 0).
inline long long d2_reset(d2_state_t& st, int src, standard_context_t& ctx) {
  hipStream_t s = ctx.stream();
  MGX_HIP(hipMemsetAsync(st.labels.data(), 0xFF, (size_t)st.n_local * sizeof(int), s));
  MGX_HIP(hipMemsetAsync(st.fs->visited.data(), 0, st.fs->visited.size() * sizeof(u32), s));
  MGX_HIP(hipMemsetAsync(st.fs->snapshot.data(), 0, st.fs->snapshot.size() * sizeof(u32), s));
  hipLaunchKernelGGL(k_d2_init, dim3(1), dim3(64), 0, s, st.args(), st.labels.data(), src, st.ranks, st.rank);
  if (src % st.ranks != st.rank) return 0;
  u64* hc = (u64*)ctx.mailbox;
  MGX_HIP(hipMemcpyAsync(hc, &st.fs->ctrl.data()->cursor[0], sizeof(u64), hipMemcpyDeviceToHost, s));
  MGX_HIP(hipStreamSynchronize(s));
  return (long long)(hc[0] & BFS_EMASK);
}

// level kernels on the local frontier (claims only), then new_bits = live & ~snapshot
inline void d2_push(d2_state_t& st, int level, standard_context_t& ctx) {
  hipStream_t s = ctx.stream();
  bfs_fused_args_t a = st.args();
  static bool attr_set = false;
  if (!attr_set) {
    MGX_HIP(hipFuncSetAttribute((const void*)k_bfs_push_level_hot<256, 4, 8192, false>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    MGX_HIP(hipFuncSetAttribute((const void*)k_bfs_push_level_wave<512, 12288>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  hipLaunchKernelGGL(k_bfs_level_begin, dim3(grid_for(st.nwords, BLOCK, 256)), dim3(BLOCK), 0, s, a, level, st.nwords);
  hipLaunchKernelGGL((k_bfs_push_level_hot<256, 4, 8192, false>), dim3(ctx.num_cus * 4), dim3(256),
                     bfs_hot_lds_bytes(256, 4, 8192), s, a, level);
  hipLaunchKernelGGL((k_bfs_push_level_wave<512, 12288>), dim3(ctx.num_cus * 2), dim3(512),
                     bfs_wave_lds_bytes(512, 12288), s, a, level);
  hipLaunchKernelGGL(k_d2_newbits, dim3(grid_for(st.nwords, BLOCK, 256)), dim3(BLOCK), 0, s, st.fs->visited.data(),
                     st.fs->snapshot.data(), st.newbits, st.nwords, a.ctrl);
}

// gathered: ranks x nwords words (every rank's new_bits).  Returns the size of this rank's next frontier;
// *new_global = vertices all ranks discovered in this level (0 on every rank at once: the traversal is over).
inline long long d2_merge(d2_state_t& st, int level, const u32* gathered, standard_context_t& ctx, long long* next_edges,
                          long long* new_global) {
  hipStream_t s = ctx.stream();
  bfs_fused_args_t a = st.args();
  hipLaunchKernelGGL(k_d2_or, dim3(grid_for(st.nwords, BLOCK, 512)), dim3(BLOCK), 0, s, gathered, st.ranks, st.nwords,
                     st.merged.data(), st.fs->visited.data(), a.ctrl);
  hipLaunchKernelGGL(k_d2_build<256>, dim3(ctx.num_cus * 4), dim3(256), 0, s, a, st.merged.data(), st.labels.data(),
                     st.n_local, st.ranks, st.rank, level);
  u64* hc = (u64*)ctx.mailbox;
  static_assert(offsetof(bfs_ctrl_t, merged_new) == 3 * sizeof(u64), "cursor[3] and merged_new are read back together");
  MGX_HIP(hipMemcpyAsync(hc, st.fs->ctrl.data(), 4 * sizeof(u64), hipMemcpyDeviceToHost, s));
  MGX_HIP(hipStreamSynchronize(s));
  const u64 cur = hc[(level + 1) % 3];
  if (new_global) *new_global = (long long)hc[3];
  if (next_edges) *next_edges = (long long)(cur & BFS_EMASK);
  return (long long)(cur >> BFS_VSHIFT);
}

}  // namespace mgx
