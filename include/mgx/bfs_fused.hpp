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

constexpr int BFS_TILE = 1024;                 // edge ranks per tile
constexpr int BFS_EPT = BFS_TILE / BLOCK;      // 4 per lane, strided by the workgroup
constexpr int BFS_STAGE = 2 * BFS_TILE;        // LDS-staged discoveries before a flush
constexpr int BFS_MAX_TRACE = 4096;            // per-level trace slots
constexpr int BFS_VSHIFT = 38;                 // cursor = (vertices << 38) | edges
constexpr u64 BFS_EMASK = (1ull << BFS_VSHIFT) - 1ull;

struct bfs_ctrl_t {
  u64 cursor[3];     // level L reads [L%3], appends into [(L+1)%3], clears [(L+2)%3]
  u64 sum_edges;     // sum over levels of E  == m_t (out-degrees of reached vertices)
  u64 sum_frontier;  // sum over levels of frontier sizes (reached vertices with degree >= 1)
  u64 reached;       // vertices labelled (incl. source and zero-degree discoveries)
  int done;
  int levels;        // number of levels that expanded at least one edge
  u64 trace[BFS_MAX_TRACE];   // cursor value each level started from
};

struct bfs_fused_args_t {
  const u32* row_offsets;
  const int* col_indices;
  int* labels;
  u32* visited;        // (n+31)/32 words
  u32* fr_row[2];      // frontier: CSR row start of each frontier vertex
  u32* fr_off[2];      // frontier: exclusive scan of degrees (edge rank of its first edge)
  bfs_ctrl_t* ctrl;
  int n;
};

__global__ void k_bfs_fused_init(bfs_fused_args_t a, int src) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  bfs_ctrl_t* c = a.ctrl;
  a.labels[src] = 0;
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
  c->done = 0;
  c->levels = 0;
}

__global__ __launch_bounds__(BLOCK) void k_bfs_push_level(bfs_fused_args_t a, int level) {
  __shared__ u32 s_off[BFS_TILE + 1];
  __shared__ u32 s_row[BFS_TILE];
  __shared__ u32 st_row[BFS_STAGE];
  __shared__ u32 st_deg[BFS_STAGE];
  __shared__ u32 s_scan[WAVES_PER_BLOCK + 1];
  __shared__ u64 s_base;
  __shared__ long long s_seg;
  __shared__ int s_count, s_wins, s_nseg;

  bfs_ctrl_t* const c = a.ctrl;
  const u64 cur = c->cursor[level % 3];
  const long long nf = (long long)(cur >> BFS_VSHIFT);
  const u64 E = cur & BFS_EMASK;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    c->cursor[(level + 2) % 3] = 0;
    if (nf == 0) {
      if (!c->done) { c->done = 1; c->levels = level; }
    } else {
      if (level < BFS_MAX_TRACE) c->trace[level] = cur;
      c->sum_edges += E;
      c->sum_frontier += (u64)nf;
    }
  }
  if (nf == 0) return;

  const u32* __restrict__ fr_row = a.fr_row[level & 1];
  const u32* __restrict__ fr_off = a.fr_off[level & 1];
  u32* __restrict__ out_row = a.fr_row[(level + 1) & 1];
  u32* __restrict__ out_off = a.fr_off[(level + 1) & 1];
  u64* const out_cursor = &c->cursor[(level + 1) % 3];

  // this workgroup's slice of edge ranks, tile aligned
  u64 per = (E + gridDim.x - 1) / gridDim.x;
  per = (per + BFS_TILE - 1) / BFS_TILE * BFS_TILE;
  const u64 e_begin = (u64)blockIdx.x * per;
  if (e_begin >= E) return;
  const u64 e_end = (e_begin + per < E) ? e_begin + per : E;

  if (threadIdx.x == 0) { s_count = 0; s_wins = 0; }
  if (threadIdx.x < WAVE) {
    const long long ub = wave_upper_bound(fr_off, nf, (u32)e_begin);
    if (threadIdx.x == 0) s_seg = ub - 1;
  }
  __syncthreads();
  long long seg = s_seg;
  const int lane = lane_id();
  const int new_label = level + 1;

  // flush the staged discoveries: one packed atomic gives frontier slots AND their degree scan
  auto flush = [&](int cnt) {
    constexpr int PER = BFS_STAGE / BLOCK;   // 8 consecutive staged entries per lane
    u32 loc[PER];      // sums stay below E < 2^32
    u32 sum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = threadIdx.x * PER + q;
      const u32 d = (i < cnt) ? st_deg[i] : 0u;
      loc[q] = sum;
      sum += d;
    }
    u32 total;
    const u32 ex = block_exclusive_sum(sum, s_scan, &total);
    if (threadIdx.x == 0) s_base = atomicAdd(out_cursor, ((u64)cnt << BFS_VSHIFT) | (u64)total);
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = threadIdx.x * PER + q;
      if (i < cnt) st_deg[i] = ex + loc[q];
    }
    __syncthreads();
    const u64 base = s_base;
    const u64 base_v = base >> BFS_VSHIFT;
    const u64 base_e = base & BFS_EMASK;
    for (int i = threadIdx.x; i < cnt; i += BLOCK) {
      out_row[base_v + i] = st_row[i];
      out_off[base_v + i] = (u32)(base_e + st_deg[i]);
    }
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
  };

  for (u64 E0 = e_begin; E0 < e_end; E0 += BFS_TILE) {
    const u64 E1 = (E0 + BFS_TILE < e_end) ? E0 + BFS_TILE : e_end;
    // stage (scanned offset, row start) of the segments beginning at `seg`, BLOCK at a time,
    // until the staged slice covers the tile.  Segments are non-empty, so BFS_TILE+1 offsets
    // always suffice; a hub row needs one round.
    int limit = 0;   // index of the last staged offset
    for (int loaded = 0;;) {
      const int j = loaded + threadIdx.x;
      if (j <= BFS_TILE) {
        const long long s = seg + j;
        s_off[j] = (s < nf) ? fr_off[s] : (u32)E;
        if (j < BFS_TILE) s_row[j] = (s < nf) ? fr_row[s] : 0u;
      }
      loaded += BLOCK;
      __syncthreads();
      limit = (loaded - 1 < BFS_TILE) ? loaded - 1 : BFS_TILE;
      if ((u64)s_off[limit] >= E1 || limit == BFS_TILE) break;
    }
    // exactly one j has s_off[j] < E1 <= s_off[j+1]
    for (int j = threadIdx.x; j < limit; j += BLOCK)
      if ((u64)s_off[j] < E1 && (u64)s_off[j + 1] >= E1) s_nseg = j + 1;
    __syncthreads();
    const int nseg = s_nseg;
    const u32 next_off = s_off[nseg];   // start of the first segment not touched by this tile

#pragma unroll
    for (int k = 0; k < BFS_EPT; ++k) {
      const u64 r = E0 + (u64)(k * BLOCK + threadIdx.x);
      bool win = false;
      u32 ro = 0, deg = 0;
      if (r < E1) {
        int lo = 0, hi = nseg;          // largest j with s_off[j] <= r
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if ((u64)s_off[mid] <= r) lo = mid; else hi = mid;
        }
        const u32 e = s_row[lo] + (u32)(r - s_off[lo]);
        const int dst = a.col_indices[e];
        const u32 bit = 1u << (dst & 31);
        u32* const word = a.visited + (dst >> 5);
        if (!(*word & bit)) {
          const u32 old = atomicOr(word, bit);
          if (!(old & bit)) {
            win = true;
            a.labels[dst] = new_label;
            ro = a.row_offsets[dst];
            deg = a.row_offsets[dst + 1] - ro;
          }
        }
      }
      const u64 mw = __ballot(win);
      if (mw) {
        const u64 ms = __ballot(win && deg != 0);
        int base = 0;
        if (lane == 0) {
          atomicAdd(&s_wins, __popcll(mw));
          if (ms) base = atomicAdd(&s_count, __popcll(ms));
        }
        base = __shfl(base, 0, WAVE);
        if (win && deg != 0) {
          const int pos = base + rank_in_mask(ms);
          st_row[pos] = ro;
          st_deg[pos] = deg;
        }
      }
    }
    __syncthreads();
    const int cnt = s_count;
    if (cnt >= BFS_TILE) flush(cnt);
    seg += ((u64)next_off == E1) ? nseg : nseg - 1;
  }
  {
    const int cnt = s_count;   // stable: last loop iteration ended with a barrier
    if (cnt > 0) flush(cnt);
  }
  if (threadIdx.x == 0 && s_wins) atomicAdd(&c->reached, (u64)s_wins);
}

struct bfs_fused_state_t {
  mem_t<u32> visited;
  mem_t<u32> fr_row[2];
  mem_t<u32> fr_off[2];
  mem_t<bfs_ctrl_t> ctrl;
  bfs_ctrl_t* host_ctrl = nullptr;   // pinned copy for stats
  int n = 0;
  int levels_per_sync = 8;
  int grid = 0;
  // timing of the level kernels of the last run (HIP events around each batch of launches)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double level_kernel_ms = 0.0;
  long long level_kernel_launches = 0;

  bfs_fused_state_t() {}
  bfs_fused_state_t(int num_nodes, standard_context_t& ctx) : n(num_nodes) {
    visited = mem_t<u32>((size_t)(num_nodes + 31) / 32 + 1, ctx);
    for (int i = 0; i < 2; ++i) {
      fr_row[i] = mem_t<u32>((size_t)num_nodes + 1, ctx);
      fr_off[i] = mem_t<u32>((size_t)num_nodes + 1, ctx);
    }
    ctrl = mem_t<bfs_ctrl_t>(1, ctx);
    MGX_HIP(hipHostMalloc((void**)&host_ctrl, sizeof(bfs_ctrl_t), hipHostMallocDefault));
    MGX_HIP(hipEventCreate(&ev0));
    MGX_HIP(hipEventCreate(&ev1));
    grid = ctx.num_cus * 6;
    if (const char* e = getenv("MGX_BFS_LEVELS_PER_SYNC")) levels_per_sync = atoi(e) > 0 ? atoi(e) : 8;
    if (const char* e = getenv("MGX_BFS_GRID")) grid = atoi(e) > 0 ? atoi(e) : grid;
  }
  bfs_fused_state_t(bfs_fused_state_t&& r) noexcept { *this = std::move(r); }
  bfs_fused_state_t& operator=(bfs_fused_state_t&& r) noexcept {
    visited = std::move(r.visited);
    for (int i = 0; i < 2; ++i) { fr_row[i] = std::move(r.fr_row[i]); fr_off[i] = std::move(r.fr_off[i]); }
    ctrl = std::move(r.ctrl);
    std::swap(host_ctrl, r.host_ctrl);
    std::swap(ev0, r.ev0);
    std::swap(ev1, r.ev1);
    n = r.n; levels_per_sync = r.levels_per_sync; grid = r.grid;
    return *this;
  }
  ~bfs_fused_state_t() {
    if (host_ctrl) (void)hipHostFree(host_ctrl);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
  }
};

// Runs a whole push BFS from `src` on the context's stream.  labels[] is (re)initialised here.
// Returns with the stream synchronised and host_ctrl holding the final counters.
inline void bfs_fused_push_run(bfs_fused_state_t& st, const int* row_offsets, const int* col_indices, int* labels,
                               int src, standard_context_t& ctx) {
  hipStream_t s = ctx.stream();
  bfs_fused_args_t a;
  a.row_offsets = (const u32*)row_offsets;
  a.col_indices = col_indices;
  a.labels = labels;
  a.visited = st.visited.data();
  for (int i = 0; i < 2; ++i) { a.fr_row[i] = st.fr_row[i].data(); a.fr_off[i] = st.fr_off[i].data(); }
  a.ctrl = st.ctrl.data();
  a.n = st.n;
  MGX_HIP(hipMemsetAsync(labels, 0xFF, (size_t)st.n * sizeof(int), s));
  MGX_HIP(hipMemsetAsync(st.visited.data(), 0, st.visited.size() * sizeof(u32), s));
  hipLaunchKernelGGL(k_bfs_fused_init, dim3(1), dim3(64), 0, s, a, src);
  int level = 0;
  st.level_kernel_ms = 0.0;
  st.level_kernel_launches = 0;
  for (;;) {
    MGX_HIP(hipEventRecord(st.ev0, s));
    for (int i = 0; i < st.levels_per_sync; ++i, ++level)
      hipLaunchKernelGGL(k_bfs_push_level, dim3(st.grid), dim3(BLOCK), 0, s, a, level);
    MGX_HIP(hipEventRecord(st.ev1, s));
    MGX_HIP(hipMemcpyAsync(&st.host_ctrl->done, &st.ctrl.data()->done, sizeof(int), hipMemcpyDeviceToHost, s));
    MGX_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    MGX_HIP(hipEventElapsedTime(&ms, st.ev0, st.ev1));
    st.level_kernel_ms += ms;
    st.level_kernel_launches += st.levels_per_sync;
    if (st.host_ctrl->done) break;
  }
  MGX_HIP(hipMemcpyAsync(st.host_ctrl, st.ctrl.data(), sizeof(bfs_ctrl_t), hipMemcpyDeviceToHost, s));
  MGX_HIP(hipStreamSynchronize(s));
}

}  // namespace mgx
