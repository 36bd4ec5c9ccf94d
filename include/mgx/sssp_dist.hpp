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
//   exchange bin sizes, then all-to-all-v of the 8-byte pairs over RCCL/xGMI: by the library itself in dsssp_run below (one
//            group of sends and receives on the context's stream), or by the caller between the per-step entry points
//            (torch.distributed in mini_amd/dist_sssp.py: the gloo tests);
//   receive  the owner takes the minimum of what arrives (atomicMin); a vertex whose distance dropped joins the next
//            frontier once (flag + list);
//   swap     next frontier becomes current; its global size (all-reduce of one int) ends the loop at zero.
// Distances are non-negative floats compared through their integer view (IEEE order), as in the single-GPU engine
// (gunrock/intrinsics.hxx); the result is the min-plus fixed point, the same bits as the single-GPU loop and the oracle.
#pragma once
#include <vector>

#include "comm.hpp"
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

// ---- the superstep loop with the library's own communicator ----------------------------------------------------------------
// What mini_amd/dist_sssp.py did with torch.distributed between ctypes calls, on the context's stream with direct RCCL
// calls: expand -> the ranks' bin counts (one all-gather of R counters per rank: everybody learns the whole R x R matrix,
// a rank reads its column) -> the all-to-all-v of the 8-byte pairs as ONE group of R - 1 sends and R - 1 receives (SURVEY
// 8e: all seven links of a GPU busy at once) -> receive -> swap -> the global frontier size (an all-gather of one counter).
// Three host waits per superstep (bin counts, matrix, frontier size): the message sizes have to be known on the host.
struct dsssp_run_bufs_t {
  mem_t<unsigned long long> matrix;     // R x R bin counts (row = sender), then R frontier sizes
  mem_t<unsigned long long> recv;       // what the other ranks send in a superstep: at most bin_cap pairs each
  mem_t<unsigned long long> mine;       // my row of counters / my frontier size, as the all-gathers' send buffer
  unsigned long long* host_matrix = nullptr;
  dsssp_run_bufs_t() {}
  dsssp_run_bufs_t(const dsssp_run_bufs_t&) = delete;
  dsssp_run_bufs_t& operator=(const dsssp_run_bufs_t&) = delete;
  ~dsssp_run_bufs_t() { if (host_matrix) (void)hipHostFree(host_matrix); }
  void ensure(const dsssp_state_t& st, standard_context_t& ctx) {
    const size_t R = (size_t)st.ranks;
    if (matrix.size() < R * R + R) {
      matrix = mem_t<unsigned long long>(R * R + R, ctx);
      mine = mem_t<unsigned long long>(R + 1, ctx);
      if (host_matrix) (void)hipHostFree(host_matrix);
      MGX_HIP(hipHostMalloc((void**)&host_matrix, (R * R + R) * sizeof(unsigned long long), hipHostMallocDefault));
    }
    const size_t want = R > 1 ? (R - 1) * (size_t)st.bin_cap + 1 : 1;
    if (recv.size() < want) recv = mem_t<unsigned long long>(want, ctx);
  }
};

// out4: supersteps, edges relaxed here, pairs sent from here, pairs received here
inline void dsssp_run(dsssp_state_t& st, comm_t& cm, dsssp_run_bufs_t& bufs, int src_global, standard_context_t& ctx, long long* out4) {
  const rccl_api_t& api = cm.table();
  const int R = st.ranks, me = st.rank;
  if (R > 1 && !(api.ok() && cm.comm)) throw mgx_error(MGX_E_INVALID, "dsssp_run: more than one rank needs a communicator");
  const bool coll = cm.comm != nullptr;          // (a one-rank communicator still runs the collectives: the tests' way to exercise them on one GPU)
  hipStream_t s = ctx.stream();
  bufs.ensure(st, ctx);
  dsssp_reset(st, src_global, ctx);
  long long supersteps = 0, relaxed = 0, sent = 0, received = 0;
  for (;;) {
    long long edges = 0;
    dsssp_expand(st, ctx, &edges);                    // (waits: host_counters[0 .. R) = this rank's bin counts)
    relaxed += edges;
    if (coll) {
      st.host_counters[me] = 0;                       // (a rank never bins its own vertices: expand relaxed them)
      MGX_HIP(hipMemcpyAsync(bufs.mine.data(), st.host_counters, (size_t)R * sizeof(unsigned long long), hipMemcpyHostToDevice, s));
      MGX_RCCL(api.AllGather(bufs.mine.data(), bufs.matrix.data(), (size_t)R, ncclUint64, cm.comm, s));
      MGX_HIP(hipMemcpyAsync(bufs.host_matrix, bufs.matrix.data(), (size_t)R * R * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
      MGX_HIP(hipStreamSynchronize(s));
      std::vector<long long> from(R, 0);
      long long total = 0;
      for (int r = 0; r < R; ++r) {
        from[r] = r == me ? 0 : (long long)bufs.host_matrix[(size_t)r * R + me];
        if (from[r] > st.bin_cap) throw mgx_error(MGX_E_INVALID, "dsssp_run: a rank announced more pairs than a bin holds");
        total += from[r];
      }
      bool any = total > 0;
      for (int r = 0; r < R; ++r) any |= r != me && st.host_counters[r] > 0;
      if (any) {
        rccl_group_t group(api);
        long long at = 0;
        for (int r = 0; r < R; ++r) {
          if (r == me) continue;
          const long long out = (long long)st.host_counters[r];
          if (out > 0) { MGX_RCCL(api.Send(st.bins.data() + (size_t)r * st.bin_cap, (size_t)out, ncclUint64, r, cm.comm, s)); sent += out; }
          if (from[r] > 0) { MGX_RCCL(api.Recv(bufs.recv.data() + at, (size_t)from[r], ncclUint64, r, cm.comm, s)); at += from[r]; }
        }
        group.end();
      }
      received += total;
      dsssp_receive(st, bufs.recv.data(), total, ctx);
    }
    long long nf = dsssp_swap(st, ctx);                // (waits)
    ++supersteps;
    if (coll) {
      bufs.host_matrix[(size_t)R * R] = (unsigned long long)nf;
      MGX_HIP(hipMemcpyAsync(bufs.mine.data() + R, bufs.host_matrix + (size_t)R * R, sizeof(unsigned long long), hipMemcpyHostToDevice, s));
      MGX_RCCL(api.AllGather(bufs.mine.data() + R, bufs.matrix.data() + (size_t)R * R, 1, ncclUint64, cm.comm, s));
      MGX_HIP(hipMemcpyAsync(bufs.host_matrix, bufs.matrix.data() + (size_t)R * R, (size_t)R * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
      MGX_HIP(hipStreamSynchronize(s));
      nf = 0;
      for (int r = 0; r < R; ++r) nf += (long long)bufs.host_matrix[r];
    }
    if (nf == 0) break;
  }
  out4[0] = supersteps; out4[1] = relaxed; out4[2] = sent; out4[3] = received;
}

}  // namespace mgx
