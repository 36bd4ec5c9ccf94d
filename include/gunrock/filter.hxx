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
  // two launches over the same ballots: the first evaluates the functor once per element and counts (the count is the
  // operator's return value: it has to reach the host anyway), the second scatters the survivors -- it reads only them
  const int* const candidates = input->data()->data();
  typename Problem::data_slice_t* const slice = problem->d_data_slice.data();
  auto pass = mgx::transform_compact((long long)input->size(), context);
  // the advance that produced this very frontier may have evaluated cond_filter per slot already (advance.hxx: functors whose test
  // looks at the slot's value alone) and left the ballots where this pass keeps its own: then the test is not run again
  // ("exactly once per input element" holds either way) and the raw frontier is read only to copy the survivors
  const mgx::standard_context_t::keep_record_t rec = context.keep;
  context.keep.valid = false;
  const bool ready = rec.valid && !input->exposed() && rec.data == (const void*)candidates && rec.n == (long long)input->size() && rec.iteration == iteration &&
                     rec.functor == (const void*)&mgx::functor_tag_t<Functor>::id && rec.epoch == context.scratch_epoch &&
                     rec.generation == mgx::frontier_generation();     // (nobody has written into a frontier from outside since)
  const long long kept = ready ? pass.upsweep_from_bits()
                               : pass.upsweep([=] __device__(long long i) { return Functor::cond_filter(candidates[i], slice, iteration); });
  output->resize((size_t)kept);
  int* const survivors = output->data()->data();
  pass.downsweep([=] __device__(long long to, long long from) { survivors[to] = candidates[from]; });
  return (int)kept;
}

// uniquify_kernel (filter.hxx:95-119): the filter of the IDEMPOTENT traversal mode -- advance<idempotence = true>
// emits every neighbour of the frontier, visited or not and with all duplicates (advance.hxx:60); uniquify removes what
// has been seen and hands the rest to ProblemFunctor::cond_uniq.  Upstream, the three heuristic culls in front of
// cond_uniq (bitmask / warp hash / history hash, :33-91) are dead code: no enactor instantiates them, bitmask_cull has an
// operator-precedence bug (`1 << item & 7`), none of them is exact (plain byte read-modify-write of the mask) and every
// one skips vertex 0.  Here is their INTENT for wave64:
//   1. intra-wave cull: of the lanes of a wave that hold the same vertex one goes on (mgx::wave_first_of_equal: LDS hash
//      + one shuffle) -- duplicates cluster, because a hub is the neighbour of many frontier vertices;
//   2. EXACT visited-bitmask cull: atomicOr on the caller's d_visited_mask, one bit per vertex ((n + 31) / 32 words);
//      exactly one edge per vertex and traversal gets past it, so what cond_uniq does to the vertex is race-free;
//   3. ProblemFunctor::cond_uniq.
// Items culled are dropped from the output (stable: survivors keep their order); the input frontier is left untouched
// (upstream overwrites culled slots with -1).  Returns nothing, like upstream: the output's size() is the count.
template <typename Problem, typename ProblemFunctor>
void uniquify_kernel(std::shared_ptr<Problem> problem, unsigned char* d_visited_mask,
                     std::shared_ptr<frontier_t<int>>& input, std::shared_ptr<frontier_t<int>>& output,
                     int iteration, standard_context_t& context) {
  const int* const candidates = input->data()->data();
  typename Problem::data_slice_t* const slice = problem->d_data_slice.data();
  unsigned* const seen = (unsigned*)d_visited_mask;   // caller allocates (n+31)/32 words
  auto pass = mgx::transform_compact((long long)input->size(), context);
  const long long kept = pass.upsweep([=] __device__(long long i) {
    const int v = candidates[i];
    if (v < 0) return false;
    const unsigned bit = 1u << (v & 31);
    if (seen[v >> 5] & bit) return false;               // in an earlier call (or earlier in this one)
    if (!mgx::wave_first_of_equal(v)) return false;     // another lane of this wave carries it on
    if (atomicOr(seen + (v >> 5), bit) & bit) return false;   // somebody else got there first
    return ProblemFunctor::cond_uniq(v, slice, iteration);
  });
  output->resize((size_t)kept);
  int* const survivors = output->data()->data();
  pass.downsweep([=] __device__(long long to, long long from) { survivors[to] = candidates[from]; });
}

}  // namespace filter
}  // namespace oprtr
}  // namespace gunrock
