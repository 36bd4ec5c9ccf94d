// gunrock/kcore/kcore_functor.hxx -- the three device functors of the k-core peeling (the C-ABI's pre-instantiated
// ones).  Arithmetic of the reference's gunrock/src/kcore/kcore_functor.hxx; `k` arrives in the operators' `iteration`
// argument.
//   deg_less_than_k_functor_t::cond_filter   a vertex that still has entries (degree > 0) but fewer than k leaves the
//                                            graph in this pass: its core number is k - 1, its degree becomes 0  (:11-19)
//   deg_atleast_k_functor_t::cond_filter     degree >= k: still in the running for the k-core                    (:22-26)
//   update_deg_functor_t                     cond_advance: always; apply_advance: the neighbour of a leaving vertex
//                                            loses one degree per entry, atomically -- a vertex that has left keeps
//                                            being decremented below 0, which is why the first functor asks for
//                                            degree > 0 as well                                                  (:28-35)
#pragma once
#include "../intrinsics.hxx"
#include "kcore_problem.hxx"

namespace gunrock {
namespace kcore {

typedef kcore_problem_t::data_slice_t kcore_slice_t;

struct deg_less_than_k_functor_t {
  static __device__ __forceinline__ bool cond_filter(int v, kcore_slice_t* d, int k) {
    const int degree = d->d_degrees[v];
    const bool leaves = degree > 0 && degree < k;
    if (leaves) {
      d->d_degrees[v] = 0;
      d->d_num_cores[v] = k - 1;
    }
    return leaves;
  }
};

struct deg_atleast_k_functor_t {
  static __device__ __forceinline__ bool cond_filter(int v, kcore_slice_t* d, int k) { return d->d_degrees[v] >= k; }
};

struct update_deg_functor_t {
  static __device__ __forceinline__ bool cond_advance(int, int, int, int, int, kcore_slice_t*, int) { return true; }
  // upstream returns `degrees[dst] > 0` read back after the add; the advance runs with has_output = false, so nobody
  // looks at it -- the value the atomic returned answers the same question without a second load
  static __device__ __forceinline__ bool apply_advance(int, int dst, int, int, int, kcore_slice_t* d, int) {
    return atomicAdd(d->d_degrees + dst, -1) > 1;
  }
};

}  // namespace kcore
}  // namespace gunrock
