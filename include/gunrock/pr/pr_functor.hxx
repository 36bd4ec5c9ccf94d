// gunrock/pr/pr_functor.hxx -- device functor of the PageRank-style iteration (the C-ABI's pre-instantiated one).
// Arithmetic of the reference's pr_functor_t (gunrock/src/pr/pr_functor.hxx:10-32):
//   cond_filter          rank <- 0.15 + 0.85 * reduced / degree (0.15 for an isolated vertex, 0 if that is not
//                        finite); the vertex stays in the frontier while its rank moved by more than 0.1 % (:11-17)
//   get_value_to_reduce  the neighbour's rank, 0 if not finite                                             (:27-29)
//   cond_advance / apply_advance   unused by the loop, always true                                         (:19-25)
#pragma once
#include <cmath>

#include "../intrinsics.hxx"
#include "pr_problem.hxx"

namespace gunrock {
namespace pr {

struct pr_functor_t {
  using slice_t = pr_problem_t::data_slice_t;
  // cond_advance / apply_advance are trivially true and get_value_to_reduce is a pure read: the operator may take the
  // value of a vertex once instead of once per edge and skip the two calls (include/gunrock/neighborhood.hxx)
  static constexpr bool mgx_pure_gather = true;

  static __device__ __forceinline__ float finite_or_zero(float x) { return isfinite(x) ? x : 0.0f; }

  static __device__ __forceinline__ float get_value_to_reduce(int v, slice_t* d, int) {
    return finite_or_zero(d->d_current_ranks[v]);
  }

  static __device__ __forceinline__ bool cond_filter(int v, slice_t* d, int) {
    const float before = d->d_current_ranks[v];
    const float degree = d->d_degrees[v];
    const float after = finite_or_zero(degree > 0 ? 0.15f + 0.85f * d->d_reduced_ranks[v] / degree : 0.15f);
    d->d_current_ranks[v] = after;
    return fabsf(after - before) > 0.001f * before;
  }

  static __device__ __forceinline__ bool cond_advance(int, int, int, int, int, slice_t*, int) { return true; }
  static __device__ __forceinline__ bool apply_advance(int, int, int, int, int, slice_t*, int) { return true; }
};

}  // namespace pr
}  // namespace gunrock
