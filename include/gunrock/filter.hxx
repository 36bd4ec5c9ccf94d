// gunrock/filter.hxx -- the filter operator: stable compaction by Functor::cond_filter.
// Drop-in for the reference's gunrock/src/filter.hxx (filter_kernel :11-31, uniquify_kernel
// :95-119).  cond_filter is evaluated exactly once per input element, output keeps input order.
#pragma once

#include "../mgx/scan.hpp"
#include "frontier.hxx"

namespace gunrock {
namespace oprtr {
namespace filter {

template <typename Problem, typename Functor>
int filter_kernel(std::shared_ptr<Problem> problem, std::shared_ptr<frontier_t<int>>& input,
                  std::shared_ptr<frontier_t<int>>& output, int iteration, standard_context_t& context) {
  auto compact = mgx::transform_compact((long long)input->size(), context);
  const int* input_data = input->data()->data();
  typename Problem::data_slice_t* data = problem->d_data_slice.data();
  const long long stream_count = compact.upsweep([=] __device__(long long idx) {
    const int item = input_data[idx];
    return Functor::cond_filter(item, data, iteration);
  });
  output->resize((size_t)stream_count);
  int* output_data = output->data()->data();
  compact.downsweep(
      [=] __device__(long long dest_idx, long long source_idx) { output_data[dest_idx] = input_data[source_idx]; });
  return (int)stream_count;
}

// uniquify_kernel (filter.hxx:95-119): the reference's heuristic culls (bitmask / warp hash /
// history hash, :33-91) are dead code upstream -- no enactor instantiates them, bitmask_cull
// has an operator-precedence bug (`1 << item & 7`) and every cull skips vertex 0.
// What is implemented is their INTENT for wave64: an exact visited-bitmask cull (atomicOr on
// the caller's d_visited_mask, one bit per vertex) followed by Functor::cond_uniq.  Items
// culled are dropped from the output; the input frontier is left untouched.
template <typename Problem, typename ProblemFunctor>
void uniquify_kernel(std::shared_ptr<Problem> problem, unsigned char* d_visited_mask,
                     std::shared_ptr<frontier_t<int>>& input, std::shared_ptr<frontier_t<int>>& output,
                     int iteration, standard_context_t& context) {
  auto compact = mgx::transform_compact((long long)input->size(), context);
  const int* input_data = input->data()->data();
  typename Problem::data_slice_t* data = problem->d_data_slice.data();
  unsigned* mask_words = (unsigned*)d_visited_mask;   // caller allocates (n+31)/32 words
  const long long stream_count = compact.upsweep([=] __device__(long long idx) {
    const int item = input_data[idx];
    if (item < 0) return false;
    const unsigned bit = 1u << (item & 31);
    const unsigned old = atomicOr(mask_words + (item >> 5), bit);
    if (old & bit) return false;   // seen it
    return ProblemFunctor::cond_uniq(item, data, iteration);
  });
  output->resize((size_t)stream_count);
  int* output_data = output->data()->data();
  compact.downsweep(
      [=] __device__(long long dest_idx, long long source_idx) { output_data[dest_idx] = input_data[source_idx]; });
}

}  // namespace filter
}  // namespace oprtr
}  // namespace gunrock
