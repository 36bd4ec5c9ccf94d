// mgx/sssp_preds.hpp -- predecessors for the fused SSSP loop, as a post-pass over the final distances.
//
// The reference keeps preds inside the relaxation (sssp_functor.hxx:31-34: `preds[dst] = src` next to the atomicMin, two
// separate stores -- racy, SURVEY F7/F11: the last writer need not be the last improver) and its test compares them with the
// CPU's exactly (tests/sssp/test_sssp.cu:44-51), which only holds by luck on a GPU.  The fused loop (sssp_fused.hpp) keeps its
// relaxation free of that second scattered store; what a caller can rely on is built here, from the fixed point alone:
//
//   pred[v] = a vertex u with an edge u -> v that is TIGHT, dist[u] + w(u, v) == dist[v] (the float sum the loop itself
//   computed), chosen so that the preds form a tree rooted at the source; pred[source] = pred[unreached] = -1.
//
// Pass 1 (every edge once, load-balanced by chunks of consecutive entries): a tight edge from a strictly nearer u is a
// candidate at once -- following such edges the distance strictly decreases, so they can never close a cycle; the largest
// candidate wins (atomicMax: the result does not depend on the order of the threads).  A tight edge between two vertices of
// EQUAL distance (weight 0, or a weight the float sum absorbs) cannot be taken blindly -- u -> v and v -> u may both be tight --
// and goes to a work list.  Rounds 2, 3, ...: a vertex that still has no pred takes a work-list edge from a vertex that had one
// BEFORE the round began (done[u] < round): every chain of preds leads to earlier rounds and from there strictly nearer --
// acyclic again.  Ends when a round assigns nothing (weights drawn from [0, 64): one or two rounds over a list of a few thousand
// edges); a list that outgrows its buffer (a graph of zero weights) is replaced by rounds over all edges.
#pragma once
#include "runtime.hpp"
#include "wave.hpp"

namespace mgx {

constexpr int PRED_CHUNK = 1024;       // consecutive CSR entries per wave

struct pred_args_t {
  const int* row_offsets;
  const int* col_indices;
  const float* weights;
  const float* dist;
  int* pred;
  unsigned short* done;       // 0: no pred yet, k >= 1: assigned in round k (1: pass 1, the source)
  int n;
  long long m;
  int src;
  u32* wl;                    // work list: (u, v) pairs
  unsigned long long* wl_count;
  unsigned long long wl_cap;  // pairs
  int* changed;
  float unreached;            // the distance of a vertex the loop never reached (FLT_MAX, sssp_problem.hxx:45)
};

// the row of entry e (rows may be empty): the last r with row_offsets[r] <= e
__device__ __forceinline__ int pred_row_of(const int* __restrict__ ro, int n, long long e) {
  int lo = 0, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if ((long long)ro[mid] <= e) lo = mid; else hi = mid;
  }
  return lo;
}

// ROUND == 1: pass 1 (strict candidates by atomicMax, ties to the work list); ROUND >= 2 with FROM_LIST == false: a tie round over
// all edges (the list overflowed)
template <bool FIRST>
__global__ __launch_bounds__(BLOCK) void k_sssp_pred_edges(pred_args_t a, int round) {
  const int lane = lane_id();
  const long long wave = ((long long)blockIdx.x * BLOCK + threadIdx.x) / WAVE;
  const long long nwaves = ((long long)gridDim.x * BLOCK) / WAVE;
  bool any = false;
  for (long long e0 = wave * PRED_CHUNK; e0 < a.m; e0 += nwaves * PRED_CHUNK) {
    int r = pred_row_of(a.row_offsets, a.n, e0);          // (wave-uniform)
    const long long e1 = e0 + PRED_CHUNK < a.m ? e0 + PRED_CHUNK : a.m;
    for (long long e = e0 + lane; e < e1; e += WAVE) {
      while (r + 1 < a.n && (long long)a.row_offsets[r + 1] <= e) ++r;
      const int u = r, v = a.col_indices[e];
      const float du = a.dist[u];
      if (du == a.unreached || v == a.src || v == u) continue;
      const float dv = a.dist[v];
      if (du + a.weights[e] != dv) continue;
      if (FIRST) {
        if (du < dv) {
          atomicMax(&a.pred[v], u);
        } else {
          const unsigned long long at = atomicAdd(a.wl_count, 1ull);
          if (at < a.wl_cap) { a.wl[2 * at] = (u32)u; a.wl[2 * at + 1] = (u32)v; }
        }
      } else if (!(du < dv)) {
        const unsigned short dvv = a.done[v], duu = a.done[u];
        if ((dvv == 0 || dvv == (unsigned short)round) && duu >= 1 && duu < (unsigned short)round) {
          atomicMax(&a.pred[v], u);
          a.done[v] = (unsigned short)round;
          any = true;
        }
      }
    }
  }
  if (!FIRST && __ballot(any) && lane == 0) *a.changed = 1;
}

// after pass 1: who has a pred (or is the source)
__global__ __launch_bounds__(BLOCK) void k_sssp_pred_mark(pred_args_t a) {
  const long long v = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (v < a.n) a.done[v] = (a.pred[v] >= 0 || v == a.src) ? 1 : 0;
}

// a tie round over the work list
__global__ __launch_bounds__(BLOCK) void k_sssp_pred_round(pred_args_t a, unsigned long long count, int round) {
  bool any = false;
  for (unsigned long long i = (unsigned long long)blockIdx.x * BLOCK + threadIdx.x; i < count; i += (unsigned long long)gridDim.x * BLOCK) {
    const int u = (int)a.wl[2 * i], v = (int)a.wl[2 * i + 1];
    const unsigned short dvv = a.done[v], duu = a.done[u];
    if ((dvv == 0 || dvv == (unsigned short)round) && duu >= 1 && duu < (unsigned short)round) {
      atomicMax(&a.pred[v], u);
      a.done[v] = (unsigned short)round;
      any = true;
    }
  }
  if (__ballot(any) && lane_id() == 0) *a.changed = 1;
}

struct sssp_pred_state_t {
  mem_t<unsigned short> done;
  mem_t<u32> wl;
  mem_t<unsigned long long> counters;     // [0] work-list count, [1] (as int) changed
  unsigned long long* host = nullptr;     // pinned copy of the two
  long long last_ties = 0;                // equal-distance tight edges pass 1 found
  int last_rounds = 0;                    // rounds the last call needed (1: pass 1 alone)
  ~sssp_pred_state_t() { if (host) (void)hipHostFree(host); }
};

// pred[] (n ints, original ids like dist[]) from the final distances.  Synchronises.
inline void sssp_build_preds(sssp_pred_state_t& st, const int* row_offsets, const int* col_indices, const float* weights, const float* dist,
                             int* pred, int n, long long m, int src, float unreached, standard_context_t& ctx) {
  hipStream_t s = ctx.stream();
  if (st.done.size() < (size_t)n + 1) st.done = mem_t<unsigned short>((size_t)n + 64, ctx);
  unsigned long long cap = (unsigned long long)m / 8ull + 4096ull;
  if (st.wl.size() < 2 * cap) st.wl = mem_t<u32>((size_t)(2 * cap), ctx);
  if (!st.counters.size()) st.counters = mem_t<unsigned long long>(2, ctx);
  if (!st.host) MGX_HIP(hipHostMalloc((void**)&st.host, 64, hipHostMallocDefault));
  pred_args_t a;
  a.row_offsets = row_offsets; a.col_indices = col_indices; a.weights = weights; a.dist = dist; a.pred = pred; a.done = st.done.data();
  a.n = n; a.m = m; a.src = src; a.wl = st.wl.data(); a.wl_count = st.counters.data(); a.wl_cap = cap;
  a.changed = (int*)(st.counters.data() + 1); a.unreached = unreached;
  MGX_HIP(hipMemsetAsync(pred, 0xFF, (size_t)n * sizeof(int), s));
  MGX_HIP(hipMemsetAsync(st.counters.data(), 0, 2 * sizeof(unsigned long long), s));
  const int grid = grid_for((m + PRED_CHUNK - 1) / PRED_CHUNK * WAVE, BLOCK, ctx.num_cus * 16);
  if (m > 0) hipLaunchKernelGGL(k_sssp_pred_edges<true>, dim3(grid), dim3(BLOCK), 0, s, a, 1);
  hipLaunchKernelGGL(k_sssp_pred_mark, dim3(grid_for(n, BLOCK, 1 << 30)), dim3(BLOCK), 0, s, a);
  MGX_CHECK_LAUNCH("SSSP predecessors: kernel launch");
  MGX_HIP(hipMemcpyAsync(st.host, st.counters.data(), 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  MGX_HIP(hipStreamSynchronize(s));
  const unsigned long long ties = st.host[0];
  st.last_ties = (long long)ties;
  st.last_rounds = 1;
  if (ties == 0) return;
  const bool from_list = ties <= cap;
  for (int round = 2; round < 65535; ++round) {
    MGX_HIP(hipMemsetAsync(a.changed, 0, sizeof(int), s));
    if (from_list) hipLaunchKernelGGL(k_sssp_pred_round, dim3(grid_for((long long)ties, BLOCK, ctx.num_cus * 8)), dim3(BLOCK), 0, s, a, ties, round);
    else hipLaunchKernelGGL(k_sssp_pred_edges<false>, dim3(grid), dim3(BLOCK), 0, s, a, round);
    MGX_CHECK_LAUNCH("SSSP predecessors: tie round launch");
    MGX_HIP(hipMemcpyAsync(st.host, st.counters.data(), 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    MGX_HIP(hipStreamSynchronize(s));
    st.last_rounds = round;
    if (*(const int*)(st.host + 1) == 0) return;
  }
  // 65 533 rounds and still assigning: a chain of equal-distance vertices longer than the round stamps count.  The preds assigned
  // so far are a forest of valid tight edges; say so rather than return a partial answer silently.
  throw mgx_error(MGX_E_INVALID, "SSSP predecessors: more than 65 533 rounds of equal-distance ties (a chain of zero-weight edges that long)");
}

}  // namespace mgx
