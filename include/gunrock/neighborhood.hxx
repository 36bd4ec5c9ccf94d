// gunrock/neighborhood.hxx -- segmented neighbourhood reduce.
// Drop-in for the reference's gunrock/src/neighborhood.hxx:12-70: same six template
// parameters (push is not deducible, callers must pass it -- SURVEY F11), same call contract:
// per edge cond_advance, apply_advance, optional output write, get_value_to_reduce(neighbor);
// reduced[] is indexed by FRONTIER POSITION (:58), identity for empty segments; returns the
// number of edges visited.
#pragma once

#include "../mgx/lbs.hpp"
#include "../mgx/scan.hpp"
#include "frontier.hxx"
#include "intrinsics.hxx"

namespace gunrock {
namespace oprtr {
namespace neighborhood {

template <typename Problem, typename Functor, typename Value, typename reduce_op, bool has_output, bool push>
int neighborhood_kernel(std::shared_ptr<Problem> problem, std::shared_ptr<frontier_t<int>>& input,
                        std::shared_ptr<frontier_t<int>>& output, Value* reduced, Value identity, int iteration,
                        standard_context_t& context) {
  const int* input_data = input->data()->data();
  problem->gslice->ensure_scanned(input->capacity(), context);
  int* scanned_offsets = problem->gslice->d_scanned_row_offsets.data();
  const int* offsets = push ? problem->gslice->d_row_offsets.data() : problem->gslice->d_col_offsets.data();

  long long non_zeros = 0;
  mgx::transform_scan(
      [=] __device__(long long idx) {
        const int v = input_data[idx];
        return offsets[v + 1] - offsets[v];
      },
      (long long)input->size(), scanned_offsets, context, &non_zeros);
  if (!non_zeros) return 0;

  const int* col_indices = push ? problem->gslice->d_col_indices.data() : problem->gslice->d_row_indices.data();
  if (has_output) output->resize((size_t)non_zeros);
  int* output_data = has_output ? output->data()->data() : nullptr;
  typename Problem::data_slice_t* data = problem->d_data_slice.data();

  auto neighborhood_reduce = [=] __device__(int idx, int seg, int rank) -> Value {
    const int v = input_data[seg];
    const int start = offsets[v];
    const int neighbor = col_indices[start + rank];
    const bool cond = Functor::cond_advance(v, neighbor, start + rank, rank, idx, data, iteration);
    const bool apply = Functor::apply_advance(v, neighbor, start + rank, rank, idx, data, iteration);
    if (has_output) output_data[idx] = (cond && apply) ? neighbor : -1;
    return Functor::get_value_to_reduce(neighbor, data, iteration);
  };
  mgx::lbs_segreduce<Value>(neighborhood_reduce, non_zeros, scanned_offsets, (long long)input->size(), reduced,
                            reduce_op(), identity, context);
  return (int)non_zeros;
}

}  // namespace neighborhood
}  // namespace oprtr
}  // namespace gunrock
