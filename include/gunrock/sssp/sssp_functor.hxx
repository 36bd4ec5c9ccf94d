// gunrock/sssp/sssp_functor.hxx -- device functor for frontier Bellman-Ford SSSP.
// Same members and semantics as the reference's sssp_functor_t
// (gunrock/src/sssp/sssp_functor.hxx:10-36):
//   cond_advance   nd = dist[src] + w[e]; old = atomicMin(dist+dst, nd); true iff nd < old (:20-29)
//   apply_advance  preds[dst] = src for EVERY expanded edge, returns true (:31-34; racy by
//                  design upstream, SURVEY F7 -- distances are what is schedule-independent)
//   cond_filter    drop -1, drop ids already stamped this iteration, else stamp (:12-18).  The
//                  reference stamps with a plain load + store, which on a 64-lane wave lets up to
//                  64 copies of one id through and overflows an m*queue_sizing frontier on skewed
//                  graphs; the stamp here is ONE atomicExch (exact dedup: |frontier| <= n, so the
//                  next advance never exceeds m work items).  Same fixed point.
// util::atomicMin is one integer atomic on gfx950 (intrinsics.hxx) instead of a CAS loop.
#pragma once
#include "../intrinsics.hxx"
#include "sssp_problem.hxx"

namespace gunrock {
namespace sssp {

struct sssp_functor_t {
  typedef sssp_problem_t::data_slice_t slice_t;

  static __device__ __forceinline__ bool cond_filter(int idx, slice_t* data, int iteration) {
    if (idx == -1) return false;
    return atomicExch(data->d_visited + idx, iteration) != iteration;
  }

  static __device__ __forceinline__ bool cond_advance(int src, int dst, int edge_id, int, int, slice_t* data, int) {
    const float new_distance = data->d_labels[src] + data->d_weights[edge_id];
    // Distances only ever decrease, so a plain (possibly stale) read is an upper bound of the
    // current value: if the candidate does not beat it, the atomic could not have succeeded either.
    // Device-scope atomics run at the memory side on MI355X (~25 G/s, tools/microbench.hip); this
    // keeps them to the relaxations that can actually improve a distance.  Same fixed point.
    if (!(new_distance < data->d_labels[dst])) return false;
    const float old_distance = gunrock::util::atomicMin(data->d_labels + dst, new_distance);
    return new_distance < old_distance;
  }

  static __device__ __forceinline__ bool apply_advance(int src, int dst, int, int, int, slice_t* data, int) {
    data->d_preds[dst] = src;
    return true;
  }
};

}  // namespace sssp
}  // namespace gunrock
