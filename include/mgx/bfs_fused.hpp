// mgx/bfs_fused.hpp -- device-resident BFS: advance + filter fused per level, no host round trip per
// level.  This header holds the shared pieces: control block, level bookkeeping, the claim helper and the
// kernel that builds the next level's queues; the traversal kernels are in bfs_fused_stream.hpp (long rows),
// bfs_fused_wave.hpp (short rows) and bfs_fused_pull.hpp (bottom-up); bfs_fused_run.hpp drives them.
//
// What the reference does per level: degree scan (K1) -> 4-byte D2H (K2) ->
// load-balanced expand writing one int per EDGE, mostly -1 (K3) -> compaction upsweep + D2H (K4)
// -> downsweep (K5); ~10 launches, 2 host syncs, 4-5 cudaMalloc/cudaFree pairs, and
// 16 B/edge + 40 B/vertex of traffic.  Here a level is
//
//   k_bfs_level_begin   bookkeeping (sizes, termination flag, TEPS numerator, direction): one thread.
//   push kernels        MARK ONLY: a neighbour that is not in the visited bitmap gets mark[v] = 1, a plain byte
//                       store.  No atomics anywhere: measured on MI355X, device-scope atomics execute at the
//                       memory side (the per-XCD L2s are not coherent with each other), drop their L2 line,
//                       and together with the re-reads of the lines they dropped ran at ~5 G/s in a level with
//                       1.7 M claims -- 0.34 ms of a 0.40 ms kernel.  Byte stores are idempotent (every writer
//                       writes the same value), merge in the write-back L2s by byte mask, and are visible to
//                       the next kernel; nobody reads them while the level runs.  The visited bitmap itself is
//                       read-only during a level, so its hot prefix can sit in LDS and the rest in L2.
//   k_bfs_build         one sweep over mark[] and the bitmap: marked and not yet visited = the level's
//                       discoveries.  Sets their bits (a wave owns the words of its 64 vertices: plain
//                       stores), writes their labels and appends them to the next level's queues in batches
//                       of thousands per workgroup (two cursor atomics per batch: ONE hot 64-bit cursor
//                       takes only ~83 M returning atomics/s).
//
// Queues: a frontier is stored as (row_start, scanned_edge_offset) pairs: a batch is appended with ONE 64-bit
// atomicAdd on a packed (vertex_count << 38 | edge_count) cursor, so the slot it gets back is at once the
// queue position and the exclusive degree scan of that position -- the scan the reference recomputes every
// level (K1) comes for free, and the next level can cut its edges into equal slices.  There are two queues per
// level: rows of at least args.long_min edges (streamed row-wise) and the rest (searched per edge rank).
// Zero-degree discoveries are labelled but never queued.
//
// Algorithmic traffic: 8 B per traversed edge (col index + visited/label probe) and 20 B per
// frontier vertex -- the figure BASELINE.md's roofline uses.
#pragma once
#include "runtime.hpp"
#include "wave.hpp"

namespace mgx {

constexpr int BFS_MAX_TRACE = 4096;            // per-level trace slots
constexpr int BFS_VSHIFT = 38;                 // cursor = (vertices << 38) | edges
constexpr u64 BFS_EMASK = (1ull << BFS_VSHIFT) - 1ull;

struct bfs_ctrl_t {
  u64 cursor[3];     // short-row queue: level L reads [L%3], level L's build fills [(L+1)%3], begin clears [(L+2)%3]
  u64 merged_new;    // partitioned BFS (bfs_dist2.hpp): vertices discovered by ALL ranks in the level just merged
  u64 lcursor[3];    // long-row queue (rows of degree >= args.long_min), same packing and rotation
  u64 sum_edges;     // sum over levels of E  == m_t (out-degrees of reached vertices)
  u64 sum_frontier;  // sum over levels of frontier sizes (reached vertices with degree >= 1)
  u64 sum_long_edges;     // the part of sum_edges / sum_frontier that went through the long-row queue
  u64 sum_long_vertices;
  u64 reached;       // vertices labelled (incl. source and zero-degree discoveries)
  u64 claims;        // mark stores issued (>= reached-1: several edges may mark the same vertex)
  u64 claims_level[64];
  u64 pull_edges;    // in-edges inspected by bottom-up levels
  int done;
  int levels;        // number of levels that expanded at least one edge
  int pull;          // direction of the level about to run (set by k_bfs_level_begin; sticky once 1)
  int push_levels;   // levels run top-down
  u64 trace[BFS_MAX_TRACE];   // (vertices << 38 | edges) of each level, both queues (kept LAST: read back up to `levels`)
};

struct bfs_fused_args_t {
  const u32* row_offsets;
  const int* col_indices;
  int* labels;
  u32* visited;        // (n+31)/32 words; written only by k_bfs_build, between levels: read-only while a level runs
  unsigned char* mark; // n bytes: set by the traversal kernels for every neighbour that may be new
  u32* frontier_bits;  // direction-optimising runs: bitmap of the level's frontier (k_bfs_build: the discoveries)
  u32* fr_row[2];      // short-row queue: CSR row start of each frontier vertex
  u32* fr_off[2];      //                  exclusive scan of degrees (edge rank of its first edge)
  u32* lq_row[2];      // long-row queue, same layout
  u32* lq_off[2];
  bfs_ctrl_t* ctrl;
  const u32* in_offsets;   // in-edges for bottom-up levels (== row_offsets/col_indices on symmetric graphs)
  const int* in_indices;
  const int* old_of_new;   // hub-first layout: original id of layout vertex v (NULL = identity)
  const int* new_of_old;   // and its inverse
  int long_min;            // rows of at least this many edges go to the long-row queue (0: no such queue)
  u32 hot_min_edges;       // a push kernel copies the hot prefix of the bitmap into LDS when its queue holds at least this many edges
  int mode;                // MGX_BFS_PUSH / MGX_BFS_DIRECTION_OPT
  float alpha;             // switch to bottom-up when unvisited < frontier_vertices * alpha (bfs_enactor.hxx:68)
  int n;
  int flags;               // diagnostics only
};

__device__ __forceinline__ void bfs_ctrl_reset(bfs_ctrl_t* c) {
  for (int i = 0; i < 3; ++i) { c->cursor[i] = 0; c->lcursor[i] = 0; }
  c->merged_new = 0;
  c->sum_edges = c->sum_frontier = c->sum_long_edges = c->sum_long_vertices = 0;
  c->reached = 1;
  c->claims = 0;
  for (int i = 0; i < 64; ++i) c->claims_level[i] = 0;
  c->pull_edges = 0;
  c->done = c->levels = c->pull = c->push_levels = 0;
}

// level-0 queue entry of the source (its row is `row`, e.g. a local row of a partition)
__device__ __forceinline__ void bfs_seed_queue(const bfs_fused_args_t& a, u32 row) {
  bfs_ctrl_t* c = a.ctrl;
  const u32 ro = a.row_offsets[row];
  const u32 deg = a.row_offsets[row + 1] - ro;
  const bool is_long = a.long_min > 0 && deg >= (u32)a.long_min;
  (is_long ? a.lq_row : a.fr_row)[0][0] = ro;
  (is_long ? a.lq_off : a.fr_off)[0][0] = 0;
  (is_long ? c->lcursor : c->cursor)[0] = deg ? ((1ull << BFS_VSHIFT) | (u64)deg) : 0ull;
}

__global__ void k_bfs_fused_init(bfs_fused_args_t a, int src) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  bfs_ctrl_reset(a.ctrl);
  a.labels[src] = 0;                                   // labels live in ORIGINAL id space
  if (a.new_of_old) src = a.new_of_old[src];           // everything else in layout space
  a.visited[src >> 5] = 1u << (src & 31);
  if (a.mode == 1) a.frontier_bits[src >> 5] = 1u << (src & 31);
  bfs_seed_queue(a, (u32)src);
}

// Runs before the traversal kernels of every level: per-level bookkeeping (termination flag, trace, TEPS
// numerator) and the direction decision.  One thread.
__global__ void k_bfs_level_begin(bfs_fused_args_t a, int level) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  bfs_ctrl_t* const c = a.ctrl;
  const u64 cur = c->cursor[level % 3];
  const u64 lcur = c->lcursor[level % 3];
  const long long nf = (long long)(cur >> BFS_VSHIFT) + (long long)(lcur >> BFS_VSHIFT);
  const u64 E = (cur & BFS_EMASK) + (lcur & BFS_EMASK);
  c->cursor[(level + 2) % 3] = 0;
  c->lcursor[(level + 2) % 3] = 0;
  if (nf == 0) {
    if (!c->done) { c->done = 1; c->levels = level; }
    return;
  }
  if (level < BFS_MAX_TRACE) c->trace[level] = ((u64)nf << BFS_VSHIFT) | E;
  c->sum_edges += E;
  c->sum_frontier += (u64)nf;
  c->sum_long_edges += lcur & BFS_EMASK;
  c->sum_long_vertices += lcur >> BFS_VSHIFT;
  if (a.mode == 1 && !c->pull) {
    const float unvisited = (float)((long long)a.n - (long long)c->reached);
    if (unvisited < (float)nf * a.alpha) c->pull = 1;      // bfs_enactor.hxx:68; never switches back (:74-112)
  }
  if (!c->pull) c->push_levels += 1;
}

// ---- a level's discoveries -> bitmap, labels, next level's queues ----------------------------------------------
// FROM_MARKS (single GPU): vertex v is new when mark[v] != 0 and its bit is not set in `visited`; the kernel sets
// the bit (and frontier_bits for direction-optimising runs).  Otherwise (partitioned runs): `bits` already holds
// the discoveries of all ranks, and this rank owns the vertices `local * ranks + rank`; rows and labels are
// addressed by the local index.  Every discovery gets label level+1; those with edges are appended to the
// short- or long-row queue of level+1.  A workgroup sweeps a contiguous range and appends in batches of up to
// BUILD_FLUSH discoveries: two cursor atomics per batch, a few hundred per level for the whole device.
constexpr int BFS_BUILD_NT = 1024;
constexpr int BFS_BUILD_FLUSH = 4096;

template <int NT, bool FROM_MARKS>
__global__ __launch_bounds__(NT) void k_bfs_build(bfs_fused_args_t a, int level, const u32* __restrict__ bits,
                                                  int* __restrict__ labels, int n_local, int ranks, int rank,
                                                  int stop_when_done) {
  constexpr int NW = NT / WAVE;
  constexpr int STAGE = BFS_BUILD_FLUSH + NT;
  constexpr int PER = STAGE / NT;
  constexpr u64 CNT1 = 1ull << 40;
  constexpr u64 DEGMASK = CNT1 - 1ull;
  static_assert(STAGE % NT == 0, "stage shape");
  __shared__ u32 st_v[STAGE];
  __shared__ u64 s_scan[NW + 1];
  __shared__ u64 s_base[2];
  __shared__ int s_count, s_found;
  bfs_ctrl_t* const c = a.ctrl;
  if (stop_when_done && c->done) return;        // (a rank of a partitioned run may be handed work with an empty frontier)
  long long per_v = ((long long)n_local + gridDim.x - 1) / gridDim.x;
  per_v = (per_v + NT - 1) / NT * NT;
  const long long i_begin = (long long)blockIdx.x * per_v;
  if (i_begin >= n_local) return;
  const long long i_end = (i_begin + per_v < n_local) ? i_begin + per_v : n_local;
  if (threadIdx.x == 0) { s_count = 0; s_found = 0; }
  __syncthreads();
  const int lane = lane_id();
  const int new_label = level + 1;
  const int* __restrict__ old_of_new = a.old_of_new;
  const u32 long_min = a.long_min > 0 ? (u32)a.long_min : 0xFFFFFFFFu;
  u64* const cur_s = &c->cursor[(level + 1) % 3];
  u64* const cur_l = &c->lcursor[(level + 1) % 3];
  u32* __restrict__ const out_row_s = a.fr_row[(level + 1) & 1];
  u32* __restrict__ const out_off_s = a.fr_off[(level + 1) & 1];
  u32* __restrict__ const out_row_l = a.lq_row[(level + 1) & 1];
  u32* __restrict__ const out_off_l = a.lq_off[(level + 1) & 1];

  auto flush = [&](int cnt) {
    u32 li[PER], ro[PER], ro1[PER];
    int lab_at[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = threadIdx.x * PER + q;
      li[q] = (i < cnt) ? st_v[i] : 0u;
      ro[q] = a.row_offsets[li[q]];
      ro1[q] = a.row_offsets[li[q] + 1];
      lab_at[q] = old_of_new ? old_of_new[li[q]] : (int)li[q];
    }
    u64 loc[PER];
    u64 sum_s = 0, sum_l = 0;
    u32 longmask = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = threadIdx.x * PER + q;
      if (i < cnt) labels[lab_at[q]] = new_label;
      const u32 deg = (i < cnt) ? ro1[q] - ro[q] : 0u;
      const bool is_long = deg >= long_min;
      if (is_long) longmask |= 1u << q;
      loc[q] = is_long ? sum_l : sum_s;
      const u64 add = deg ? (CNT1 | (u64)deg) : 0ull;
      if (is_long) sum_l += add; else sum_s += add;
    }
    u64 tot_s, tot_l;
    const u64 ex_s = block_exclusive_sum_nw<NW>(sum_s, s_scan, &tot_s);
    const u64 ex_l = block_exclusive_sum_nw<NW>(sum_l, s_scan, &tot_l);
    if (threadIdx.x == 0) {
      s_base[0] = (tot_s >> 40) ? atomicAdd(cur_s, ((tot_s >> 40) << BFS_VSHIFT) | (tot_s & DEGMASK)) : 0ull;
      s_base[1] = (tot_l >> 40) ? atomicAdd(cur_l, ((tot_l >> 40) << BFS_VSHIFT) | (tot_l & DEGMASK)) : 0ull;
      s_found += cnt;
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = threadIdx.x * PER + q;
      if (i < cnt && ro1[q] != ro[q]) {
        const bool is_long = (longmask >> q) & 1u;
        const u64 base = is_long ? s_base[1] : s_base[0];
        const u64 at = (is_long ? ex_l : ex_s) + loc[q];
        const u64 slot = (base >> BFS_VSHIFT) + (at >> 40);
        (is_long ? out_row_l : out_row_s)[slot] = ro[q];
        (is_long ? out_off_l : out_off_s)[slot] = (u32)((base & BFS_EMASK) + (at & DEGMASK));
      }
    }
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
  };

  const bool want_frontier = FROM_MARKS && a.mode == 1;
  for (long long base = i_begin; base < i_end; base += NT) {          // i_begin is a multiple of NT (and of 64)
    const long long i = base + threadIdx.x;
    bool found = false;
    u32 word = 0;
    if (i < i_end) {
      if (FROM_MARKS) {
        word = a.visited[i >> 5];
        found = a.mark[i] != 0 && !((word >> (i & 31)) & 1u);
      } else {
        const long long v = i * ranks + rank;
        found = (bits[v >> 5] >> (v & 31)) & 1u;
      }
    }
    const u64 bal = __ballot(found);
    if (FROM_MARKS && i < i_end && (lane & 31) == 0) {
      // this wave is the only writer of the two words of its 64 vertices
      const u32 nb = (u32)(bal >> lane);
      if (nb) a.visited[i >> 5] = word | nb;
      if (want_frontier) a.frontier_bits[i >> 5] = nb;
    }
    const int nfound = __popcll(bal);
    if (nfound) {
      int at = 0;
      if (lane == 0) at = atomicAdd(&s_count, nfound);
      at = __builtin_amdgcn_readfirstlane(at);
      if (found) st_v[at + rank_in_mask(bal)] = (u32)i;
    }
    __syncthreads();
    const int cnt = s_count;
    if (cnt >= BFS_BUILD_FLUSH) flush(cnt);
  }
  {
    const int cnt = s_count;
    if (cnt > 0) flush(cnt);
  }
  if (threadIdx.x == 0 && s_found) atomicAdd(&c->reached, (u64)s_found);
}

// per-BFS device state of the fused engine
struct bfs_fused_state_t {
  mem_t<u32> visited;
  mem_t<unsigned char> mark;
  mem_t<u32> frontier_bits;
  mem_t<u32> fr_row[2];
  mem_t<u32> fr_off[2];
  mem_t<u32> lq_row[2];
  mem_t<u32> lq_off[2];
  mem_t<bfs_ctrl_t> ctrl;
  bfs_ctrl_t* host_ctrl = nullptr;   // pinned copy for stats
  int n = 0;
  int levels_per_sync = 8;
  int long_min = 64;                 // rows at least this long go to the long-row queue (0: no such queue)
  unsigned hot_min_edges = 65536;    // smaller levels probe the bitmap in L2 instead of copying its hot prefix to LDS
  // timing of the level kernels of the last run (HIP events around each batch of launches)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double level_kernel_ms = 0.0;
  long long level_kernel_launches = 0;
  float batch_ms[256];               // duration of each launch batch of the last run (per level when levels_per_sync == 1)
  int batches = 0;
  // per-launch timing of the two push kernels of a level: events [3i] stream [3i+1] wave [3i+2]
  static constexpr int EV_POOL = 96;
  hipEvent_t wev[EV_POOL] = {};
  double wave_kernel_ms = 0.0;       // k_bfs_push_level_wave: per-edge search over the short-row queue
  long long wave_kernel_launches = 0;
  double stream_kernel_ms = 0.0;     // k_bfs_push_level_stream: row-wise streaming of the long-row queue
  long long stream_kernel_launches = 0;
  float level_stream_ms[64] = {};    // the same per level (first 64 levels)
  float level_wave_ms[64] = {};

  bfs_fused_state_t(int num_nodes, standard_context_t& ctx) : n(num_nodes) {
    size_t words = (size_t)(num_nodes + 31) / 32 + 1;
    if (words < 65536) words = 65536;              // the kernels copy a fixed-size prefix of the bitmap into LDS
    visited = mem_t<u32>(words, ctx);
    frontier_bits = mem_t<u32>(words, ctx);
    mark = mem_t<unsigned char>((size_t)num_nodes + 64, ctx);
    MGX_HIP(hipMemsetAsync(visited.data(), 0, words * sizeof(u32), ctx.stream()));
    MGX_HIP(hipMemsetAsync(frontier_bits.data(), 0, words * sizeof(u32), ctx.stream()));
    for (int i = 0; i < 2; ++i) {
      fr_row[i] = mem_t<u32>((size_t)num_nodes + 1, ctx);
      fr_off[i] = mem_t<u32>((size_t)num_nodes + 1, ctx);
      lq_row[i] = mem_t<u32>((size_t)num_nodes + 1, ctx);
      lq_off[i] = mem_t<u32>((size_t)num_nodes + 1, ctx);
    }
    ctrl = mem_t<bfs_ctrl_t>(1, ctx);
    MGX_HIP(hipHostMalloc((void**)&host_ctrl, sizeof(bfs_ctrl_t), hipHostMallocDefault));
    MGX_HIP(hipEventCreate(&ev0));
    MGX_HIP(hipEventCreate(&ev1));
    for (int i = 0; i < EV_POOL; ++i) MGX_HIP(hipEventCreate(&wev[i]));
    if (const char* e = getenv("MGX_BFS_LONG_MIN")) long_min = atoi(e) > 0 ? atoi(e) : 0;
    if (const char* e = getenv("MGX_BFS_HOT_MIN_EDGES")) hot_min_edges = (unsigned)atoll(e);
    if (const char* e = getenv("MGX_BFS_LEVELS_PER_SYNC")) levels_per_sync = atoi(e) > 0 ? atoi(e) : 8;
  }
  bfs_fused_state_t(const bfs_fused_state_t&) = delete;
  bfs_fused_state_t& operator=(const bfs_fused_state_t&) = delete;
  ~bfs_fused_state_t() {
    if (host_ctrl) (void)hipHostFree(host_ctrl);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    for (int i = 0; i < EV_POOL; ++i) if (wev[i]) (void)hipEventDestroy(wev[i]);
  }
  size_t bitmap_words() const { return visited.size(); }
};

}  // namespace mgx
