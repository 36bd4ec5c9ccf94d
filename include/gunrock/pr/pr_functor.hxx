// gunrock/pr/pr_functor.hxx -- device functor of the PageRank-style iteration.
// Same members and arithmetic as the reference's pr_functor_t (gunrock/src/pr/pr_functor.hxx:10-32):
//   cond_filter   new = deg>0 ? 0.15 + 0.85*reduced[idx]/deg : 0.15 (non-finite -> 0); writes it;
//                 keeps the vertex while it moved by more than 0.1 % of the old value     (:11-17)
//   cond_advance / apply_advance   always true                                            (:19-25)
//   get_value_to_reduce            the neighbour's current rank (non-finite -> 0)         (:27-29)
#pragma once
#include <cmath>

#include "../intrinsics.hxx"
#include "pr_problem.hxx"

namespace gunrock {
namespace pr {

struct pr_functor_t {
  typedef pr_problem_t::data_slice_t slice_t;

  static __device__ __forceinline__ bool cond_filter(int idx, slice_t* data, int) {
    const float old_value = data->d_current_ranks[idx];
    const float deg = data->d_degrees[idx];
    float new_value = (deg > 0) ? (0.15f + 0.85f * data->d_reduced_ranks[idx] / deg) : 0.15f;
    if (!isfinite(new_value)) new_value = 0;
    data->d_current_ranks[idx] = new_value;
    return fabsf(new_value - old_value) > (0.001f * old_value);
  }
  static __device__ __forceinline__ bool cond_advance(int, int, int, int, int, slice_t*, int) { return true; }
  static __device__ __forceinline__ bool apply_advance(int, int, int, int, int, slice_t*, int) { return true; }
  static __device__ __forceinline__ float get_value_to_reduce(int idx, slice_t* data, int) {
    const float r = data->d_current_ranks[idx];
    return isfinite(r) ? r : 0.0f;
  }
};

}  // namespace pr
}  // namespace gunrock
