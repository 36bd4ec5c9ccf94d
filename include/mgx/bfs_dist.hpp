// mgx/bfs_dist.hpp -- per-rank kernels of the vertex-range partitioned BFS.
//
// The reference has no multi-GPU path at all (README.md:4); this is new design.
// Partition: rank r owns the global vertex ids [v_lo, v_hi), their CSR rows (row_offsets is local,
// n_local+1 entries; col_indices hold GLOBAL ids) and the labels of that range.
//
// One superstep on a rank:
//   expand   : load-balanced expansion of the local frontier (same scan + LBS kernels as the
//              advance operator).  Every neighbour id g goes through a rank-private `seen` bitmap
//              over ALL n_global vertices: the first edge that reaches g claims the bit and appends g
//              to the bin of g's owner; every later edge to g -- this level or any later one -- is
//              dropped locally.  So a rank sends a vertex to its owner at most ONCE per traversal:
//              the pre-send dedup that keeps the xGMI volume at O(n) instead of O(m).
//   exchange : host side (torch.distributed all_to_all over RCCL/xGMI): bin r of every rank -> rank r.
//   receive  : the owner labels the ids it has not labelled yet (CAS -1 -> level+1) and appends
//              them to its next local frontier.  The rank's own bin takes the same path without
//              leaving the device.
// Labels are the global BFS depths: identical to the single-GPU result bit for bit.
#pragma once
#include "lbs.hpp"
#include "runtime.hpp"
#include "scan.hpp"
#include "wave.hpp"

namespace mgx {

constexpr int DBFS_MAX_RANKS = 64;

struct dbfs_state_t {
  // partition
  int n_global = 0, v_lo = 0, v_hi = 0, n_local = 0, ranks = 1, rank = 0, chunk = 0;
  // graph slice (borrowed)
  const int* row_offsets = nullptr;   // n_local + 1
  const int* col_indices = nullptr;   // global ids
  long long m_local = 0;
  // state
  mem_t<int> labels;        // n_local
  mem_t<u32> seen;          // n_global bits
  mem_t<int> frontier[2];   // local row ids, capacity n_local
  mem_t<int> scanned;       // n_local + 1
  mem_t<int> bins;          // ranks * bin_cap global ids
  mem_t<unsigned long long> counters;   // [0..ranks) bin counts, [ranks] next-frontier cursor
  long long bin_cap = 0;
  long long frontier_size = 0;
  int cur = 0;
  unsigned long long* host_counters = nullptr;

  dbfs_state_t() {}
  dbfs_state_t(const dbfs_state_t&) = delete;
  dbfs_state_t& operator=(const dbfs_state_t&) = delete;
  ~dbfs_state_t() { if (host_counters) (void)hipHostFree(host_counters); }

  void init(standard_context_t& ctx, int n_global_, int v_lo_, int v_hi_, int ranks_, int rank_,
            const int* ro, const int* ci, long long m_local_, int* borrowed_bins, long long borrowed_bin_cap) {
    n_global = n_global_; v_lo = v_lo_; v_hi = v_hi_; n_local = v_hi_ - v_lo_; ranks = ranks_; rank = rank_;
    chunk = (n_global + ranks - 1) / ranks;
    row_offsets = ro; col_indices = ci; m_local = m_local_;
    labels = mem_t<int>((size_t)n_local + 1, ctx);
    seen = mem_t<u32>((size_t)(n_global + 31) / 32 + 1, ctx);
    frontier[0] = mem_t<int>((size_t)n_local + 1, ctx);
    frontier[1] = mem_t<int>((size_t)n_local + 1, ctx);
    scanned = mem_t<int>((size_t)n_local + 2, ctx);
    if (borrowed_bins) {
      bin_cap = borrowed_bin_cap;
      bins = mem_t<int>::borrow(borrowed_bins, (size_t)ranks * (size_t)bin_cap);
    } else {
      bin_cap = chunk;
      bins = mem_t<int>((size_t)ranks * (size_t)bin_cap + 1, ctx);
    }
    counters = mem_t<unsigned long long>((size_t)ranks + 1, ctx);
    MGX_HIP(hipHostMalloc((void**)&host_counters, (ranks + 1) * sizeof(unsigned long long), hipHostMallocDefault));
    ctx.reserve_scratch(scan_scratch_bytes(n_local) + (1 << 16));
  }
};

// owner of a global id under the equal-chunk range partition
__device__ __forceinline__ int dbfs_owner(int g, int chunk) { return g / chunk; }

inline void dbfs_reset(dbfs_state_t& st, int src_global, standard_context_t& ctx) {
  hipStream_t s = ctx.stream();
  MGX_HIP(hipMemsetAsync(st.labels.data(), 0xFF, (size_t)st.n_local * sizeof(int), s));
  MGX_HIP(hipMemsetAsync(st.seen.data(), 0, st.seen.size() * sizeof(u32), s));
  MGX_HIP(hipMemsetAsync(st.counters.data(), 0, st.counters.size() * sizeof(unsigned long long), s));
  st.cur = 0;
  st.frontier_size = 0;
  u32* seen = st.seen.data();
  int* labels = st.labels.data();
  int* fr = st.frontier[0].data();
  const int lo = st.v_lo, hi = st.v_hi;
  // every rank marks the source as seen (nobody must ever send it); its owner labels it
  transform(
      [=] __device__(int) {
        seen[src_global >> 5] = 1u << (src_global & 31);
        if (src_global >= lo && src_global < hi) { labels[src_global - lo] = 0; fr[0] = src_global - lo; }
      },
      1, ctx);
  if (src_global >= lo && src_global < hi) st.frontier_size = 1;
}

// expand the local frontier into per-owner bins; returns bin counts (host) via st.host_counters.
// edges_out = number of edges expanded this superstep on this rank.
inline void dbfs_expand(dbfs_state_t& st, standard_context_t& ctx, long long* edges_out) {
  hipStream_t s = ctx.stream();
  MGX_HIP(hipMemsetAsync(st.counters.data(), 0, (size_t)st.ranks * sizeof(unsigned long long), s));
  long long front = 0;
  const int* fr = st.frontier[st.cur].data();
  const int* ro = st.row_offsets;
  if (st.frontier_size > 0) {
    transform_scan(
        [=] __device__(long long i) {
          const int v = fr[i];
          return ro[v + 1] - ro[v];
        },
        st.frontier_size, st.scanned.data(), ctx, &front);
  }
  if (front > 0) {
    const int* ci = st.col_indices;
    u32* seen = st.seen.data();
    int* bins = st.bins.data();
    unsigned long long* cnt = st.counters.data();
    const int chunk = st.chunk, ranks = st.ranks;
    const long long cap = st.bin_cap;
    transform_lbs(
        [=] __device__(int idx, int seg, int rank_in_row) {
          (void)idx;
          const int v = fr[seg];
          const int g = ci[ro[v] + rank_in_row];
          const u32 bit = 1u << (g & 31);
          bool win = false;
          if (!(seen[g >> 5] & bit)) win = !(atomicOr(seen + (g >> 5), bit) & bit);
          // wave-aggregated append, one pass per destination that has a winner in this wave
          const int owner = win ? dbfs_owner(g, chunk) : -1;
          u64 pending = __ballot(win);
          while (pending) {
            const int leader = __ffsll((long long)pending) - 1;
            const int o = __shfl(owner, leader, WAVE);
            const u64 same = __ballot(win && owner == o);
            unsigned long long base = 0;
            if (lane_id() == leader) base = atomicAdd(cnt + o, (unsigned long long)__popcll(same));
            base = __shfl(base, leader, WAVE);
            if (win && owner == o) {
              const long long at = (long long)base + rank_in_mask(same);
              if (at < cap) bins[(long long)o * cap + at] = g;
            }
            pending &= ~same;
          }
          (void)ranks;
        },
        front, st.scanned.data(), st.frontier_size, ctx);
  }
  MGX_HIP(hipMemcpyAsync(st.host_counters, st.counters.data(), (size_t)st.ranks * sizeof(unsigned long long),
                         hipMemcpyDeviceToHost, s));
  MGX_HIP(hipStreamSynchronize(s));
  if (edges_out) *edges_out = front;
}

// label the not-yet-labelled ids of `ids` (global ids owned by this rank) with `label` and append
// them to the NEXT local frontier.
inline void dbfs_receive(dbfs_state_t& st, const int* ids, long long count, int label, standard_context_t& ctx) {
  if (count <= 0) return;
  int* labels = st.labels.data();
  int* next = st.frontier[st.cur ^ 1].data();
  unsigned long long* cursor = st.counters.data() + st.ranks;
  const int lo = st.v_lo;
  transform(
      [=] __device__(int i) {
        const int v = ids[i] - lo;
        bool win = false;
        if (labels[v] == -1) win = (atomicCAS(labels + v, -1, label) == -1);
        const u64 m = __ballot(win);
        if (m) {
          const int leader = __ffsll((long long)m) - 1;
          unsigned long long base = 0;
          if (lane_id() == leader) base = atomicAdd(cursor, (unsigned long long)__popcll(m));
          base = __shfl(base, leader, WAVE);
          if (win) next[base + rank_in_mask(m)] = v;
        }
      },
      count, ctx);
}

// next frontier becomes current; returns its size
inline long long dbfs_swap(dbfs_state_t& st, standard_context_t& ctx) {
  hipStream_t s = ctx.stream();
  MGX_HIP(hipMemcpyAsync(st.host_counters + st.ranks, st.counters.data() + st.ranks, sizeof(unsigned long long),
                         hipMemcpyDeviceToHost, s));
  MGX_HIP(hipMemsetAsync(st.counters.data() + st.ranks, 0, sizeof(unsigned long long), s));
  MGX_HIP(hipStreamSynchronize(s));
  st.frontier_size = (long long)st.host_counters[st.ranks];
  st.cur ^= 1;
  return st.frontier_size;
}

}  // namespace mgx
