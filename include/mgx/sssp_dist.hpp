// mgx/sssp_dist.hpp -- per-rank kernels of the vertex-range partitioned SSSP (last bullets).
//
// The reference has no multi-GPU path (README.md:4); its SSSP (sssp_enactor.hxx:40-72) is frontier Bellman-Ford: relax
// every edge of the frontier with atomicMin, the vertices whose distance improved are the next frontier.  Partitioned:
// rank r owns the global ids [v_lo, v_hi) -- their CSR rows (local row_offsets, GLOBAL col_indices, weights) and their
// distances.  One superstep on a rank:
//   expand   every edge (u -> v, w) of the local frontier gives the candidate dist[u] + w.  v local: atomicMin on the
//            owner's distance right away.  v remote: atomicMin into a rank-private array best[] over ALL vertices -- the
//            smallest candidate this rank has ever produced for v; only a candidate that lowers it is worth sending, and
//            the first one of a superstep puts v on the list of its owner's bin.  When the expansion is done a second
//            kernel turns every listed v into ONE pair (v, best[v]): MIN-COMBINING BEFORE SEND -- per destination vertex
//            and superstep a rank sends at most one pair, carrying the minimum over all its edges to v so far;
//   exchange host side (torch.distributed): bin sizes, then all-to-all-v of the 8-byte pairs over RCCL/xGMI;
//   receive  the owner takes the minimum of what arrives (atomicMin); a vertex whose distance dropped joins the next
//            frontier once (flag + list);
//   swap     next frontier becomes current; its global size (all-reduce of one int) ends the loop at zero.
// Distances are non-negative floats compared through their integer view (IEEE order), as in the single-GPU engine
// (gunrock/intrinsics.hxx); the result is the min-plus fixed point, the same bits as the single-GPU loop and the oracle.
#pragma once
#include "lbs.hpp"
#include "runtime.hpp"
#include "scan.hpp"
#include "wave.hpp"

namespace mgx {

constexpr u32 DSSSP_INF = 0x7F7FFFFFu;          // FLT_MAX: what the reference reports for unreachable vertices

struct dsssp_state_t {
  int n_global = 0, v_lo = 0, v_hi = 0, n_local = 0, ranks = 1, rank = 0, chunk = 0;
  const int* row_offsets = nullptr;   // n_local + 1 (borrowed)
  const int* col_indices = nullptr;   // global ids
  const float* weights = nullptr;
  long long m_local = 0;
  mem_t<u32> dist;             // n_local: float bits
  mem_t<u32> best;             // n_global: smallest candidate produced here for every vertex (float bits)
  mem_t<u32> listed;           // n_global: v is on a send list of the current superstep
  mem_t<u32> queued;           // n_local: v is on the next frontier
  mem_t<int> frontier[2];      // local row ids
  mem_t<int> scanned;
  mem_t<int> send_ids;         // ranks * bin_cap: the listed vertices per owner
  mem_t<unsigned long long> bins;   // ranks * bin_cap pairs (v << 32 | float bits)
  mem_t<unsigned long long> counters;   // [0..ranks) bin counts, [ranks] next-frontier cursor
  long long bin_cap = 0, frontier_size = 0;
  int cur = 0;
  unsigned long long* host_counters = nullptr;

  dsssp_state_t() {}
  dsssp_state_t(const dsssp_state_t&) = delete;
  dsssp_state_t& operator=(const dsssp_state_t&) = delete;
  ~dsssp_state_t() { if (host_counters) (void)hipHostFree(host_counters); }

  void init(standard_context_t& ctx, int n_global_, int v_lo_, int v_hi_, int ranks_, int rank_, const int* ro, const int* ci,
            const float* w, long long m_local_) {
    n_global = n_global_; v_lo = v_lo_; v_hi = v_hi_; n_local = v_hi_ - v_lo_; ranks = ranks_; rank = rank_;
    chunk = (n_global + ranks - 1) / ranks;
    row_offsets = ro; col_indices = ci; weights = w; m_local = m_local_;
    dist = mem_t<u32>((size_t)n_local + 1, ctx);
    best = mem_t<u32>((size_t)n_global + 1, ctx);
    listed = mem_t<u32>((size_t)n_global + 1, ctx);
    queued = mem_t<u32>((size_t)n_local + 1, ctx);
    frontier[0] = mem_t<int>((size_t)n_local + 1, ctx);
    frontier[1] = mem_t<int>((size_t)n_local + 1, ctx);
    scanned = mem_t<int>((size_t)n_local + 2, ctx);
    bin_cap = chunk;
    send_ids = mem_t<int>((size_t)ranks * (size_t)bin_cap + 1, ctx);
    bins = mem_t<unsigned long long>((size_t)ranks * (size_t)bin_cap + 1, ctx);
    counters = mem_t<unsigned long long>((size_t)ranks + 1, ctx);
    MGX_HIP(hipHostMalloc((void**)&host_counters, (ranks + 1) * sizeof(unsigned long long), hipHostMallocDefault));
    ctx.reserve_scratch(scan_scratch_bytes(n_local) + (1 << 16));
  }
};

inline void dsssp_reset(dsssp_state_t& st, int src_global, standard_context_t& ctx) {
  hipStream_t s = ctx.stream();
  u32* dist = st.dist.data();
  u32* best = st.best.data();
  const long long nl = st.n_local, ng = st.n_global;
  transform([=] __device__(int i) { if (i < nl) dist[i] = DSSSP_INF; best[i] = DSSSP_INF; }, nl > ng ? nl : ng, ctx);
  MGX_HIP(hipMemsetAsync(st.listed.data(), 0, st.listed.size() * sizeof(u32), s));
  MGX_HIP(hipMemsetAsync(st.queued.data(), 0, st.queued.size() * sizeof(u32), s));
  MGX_HIP(hipMemsetAsync(st.counters.data(), 0, st.counters.size() * sizeof(unsigned long long), s));
  st.cur = 0;
  st.frontier_size = 0;
  const int lo = st.v_lo, hi = st.v_hi;
  int* fr = st.frontier[0].data();
  // nobody ever needs to send the source anything: its distance is final
  transform([=] __device__(int) {
    best[src_global] = 0u;
    if (src_global >= lo && src_global < hi) { dist[src_global - lo] = 0u; fr[0] = src_global - lo; }
  }, 1, ctx);
  if (src_global >= lo && src_global < hi) st.frontier_size = 1;
}

// wave-aggregated append of `item` for the lanes with `want` set, grouped by `key` (a bin number): one returning
// atomic per (wave, key)
template <typename T>
__device__ __forceinline__ void dsssp_append(bool want, int key, T item, T* base, long long cap, unsigned long long* counts) {
  u64 pending = __ballot(want);
  while (pending) {
    const int leader = __ffsll((long long)pending) - 1;
    const int k = __shfl(key, leader, WAVE);
    const u64 same = __ballot(want && key == k);
    unsigned long long at0 = 0;
    if (lane_id() == leader) at0 = atomicAdd(counts + k, (unsigned long long)__popcll(same));
    at0 = __shfl(at0, leader, WAVE);
    if (want && key == k) {
      const long long at = (long long)at0 + rank_in_mask(same);
      if (at < cap) base[(long long)k * cap + at] = item;
    }
    pending &= ~same;
  }
}

// Relax every edge of the local frontier; bins of (v, best[v]) pairs per owner; counts in st.host_counters.
inline void dsssp_expand(dsssp_state_t& st, standard_context_t& ctx, long long* edges_out) {
  hipStream_t s = ctx.stream();
  MGX_HIP(hipMemsetAsync(st.counters.data(), 0, (size_t)st.ranks * sizeof(unsigned long long), s));
  long long front = 0;
  const int* fr = st.frontier[st.cur].data();
  const int* ro = st.row_offsets;
  if (st.frontier_size > 0)
    transform_scan([=] __device__(long long i) { const int v = fr[i]; return ro[v + 1] - ro[v]; }, st.frontier_size,
                   st.scanned.data(), ctx, &front);
  const int ranks = st.ranks, chunk = st.chunk, lo = st.v_lo, hi = st.v_hi;
  const long long cap = st.bin_cap;
  unsigned long long* cnt = st.counters.data();
  u32* listed = st.listed.data();
  u32* best = st.best.data();
  int* send_ids = st.send_ids.data();
  if (st.frontier_size > 0) {
    u32* queued = st.queued.data();
    // leaving the queue: a vertex whose distance drops again during this superstep is queued again
    transform([=] __device__(int i) { queued[fr[i]] = 0u; }, st.frontier_size, ctx);
  }
  if (front > 0) {
    const int* ci = st.col_indices;
    const float* w = st.weights;
    u32* dist = st.dist.data();
    u32* queued = st.queued.data();
    int* next = st.frontier[st.cur ^ 1].data();
    unsigned long long* next_cursor = cnt + ranks;
    transform_lbs(
        [=] __device__(int idx, int seg, int rank_in_row) {
          (void)idx;
          const int u = fr[seg];
          const int e = ro[u] + rank_in_row;
          const int g = ci[e];
          const u32 cand = __float_as_uint(__uint_as_float(dist[u]) + w[e]);     // non-negative floats: integer order
          bool to_next = false, to_send = false;
          if (g >= lo && g < hi) {
            if (cand < dist[g - lo] && cand < atomicMin(dist + (g - lo), cand)) to_next = atomicExch(queued + (g - lo), 1u) == 0u;
          } else {
            if (cand < best[g] && cand < atomicMin(best + g, cand)) to_send = atomicExch(listed + g, 1u) == 0u;
          }
          dsssp_append(to_next, 0, g - lo, next, (long long)0x7FFFFFFF, next_cursor);
          dsssp_append(to_send, g / chunk, g, send_ids, cap, cnt);
        },
        front, st.scanned.data(), st.frontier_size, ctx);
    // one pair per listed vertex: the minimum over everything this rank has found for it
    unsigned long long* bins = st.bins.data();
    transform(
        [=] __device__(int i) {
          const int r = i / (int)cap, k = i - r * (int)cap;
          if ((unsigned long long)k < cnt[r]) {
            const int g = send_ids[(long long)r * cap + k];
            listed[g] = 0u;
            bins[(long long)r * cap + k] = ((unsigned long long)(u32)g << 32) | (unsigned long long)best[g];
          }
        },
        (long long)ranks * cap, ctx);
  }
  MGX_HIP(hipMemcpyAsync(st.host_counters, cnt, (size_t)ranks * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  MGX_HIP(hipStreamSynchronize(s));
  if (edges_out) *edges_out = front;
}

// pairs (v << 32 | float bits) for vertices this rank owns: keep the minimum; improved vertices join the next frontier
inline void dsssp_receive(dsssp_state_t& st, const unsigned long long* pairs, long long count, standard_context_t& ctx) {
  if (count <= 0) return;
  u32* dist = st.dist.data();
  u32* queued = st.queued.data();
  int* next = st.frontier[st.cur ^ 1].data();
  unsigned long long* next_cursor = st.counters.data() + st.ranks;
  const int lo = st.v_lo;
  transform(
      [=] __device__(int i) {
        const unsigned long long p = pairs[i];
        const int v = (int)(p >> 32) - lo;
        const u32 d = (u32)p;
        bool to_next = false;
        if (d < dist[v] && d < atomicMin(dist + v, d)) to_next = atomicExch(queued + v, 1u) == 0u;
        dsssp_append(to_next, 0, v, next, (long long)0x7FFFFFFF, next_cursor);
      },
      count, ctx);
}

// next frontier becomes current; returns its size
inline long long dsssp_swap(dsssp_state_t& st, standard_context_t& ctx) {
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
