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
  graph_device_t& graph = *problem->gslice;
  // out-edges (CSR) when pushing, in-edges (the CSC slots) when pulling
  const int* const offsets = push ? graph.d_row_offsets.data() : graph.d_col_offsets.data();
  const int* const neighbours = push ? graph.d_col_indices.data() : graph.d_row_indices.data();
  const int* const frontier = input->data()->data();
  const long long frontier_size = (long long)input->size();

  // segment i = the neighbour list of frontier[i]: exclusive scan of the degrees into the graph's scratch scan
  graph.ensure_scanned(input->capacity(), context);
  int* const segment_start = graph.d_scanned_row_offsets.data();
  long long edges = 0;
  mgx::transform_scan([=] __device__(long long i) { return offsets[frontier[i] + 1] - offsets[frontier[i]]; },
                      frontier_size, segment_start, context, &edges);
  if (edges == 0) return 0;

  int* out = nullptr;
  if (has_output) {
    output->resize((size_t)edges);
    out = output->data()->data();
  }
  typename Problem::data_slice_t* const data = problem->d_data_slice.data();

  // per edge: the functor's contract (cond, then apply, both always), the optional output slot, and the value that
  // goes into the segment's reduction; reduced[i] belongs to frontier POSITION i (neighborhood.hxx:58)
  auto visit = [=] __device__(int slot, int segment, int rank) -> Value {
    const int v = frontier[segment];
    const int edge = offsets[v] + rank;
    const int u = neighbours[edge];
    const bool cond = Functor::cond_advance(v, u, edge, rank, slot, data, iteration);
    const bool applied = Functor::apply_advance(v, u, edge, rank, slot, data, iteration);
    if (has_output) out[slot] = (cond && applied) ? u : -1;
    return Functor::get_value_to_reduce(u, data, iteration);
  };
  mgx::lbs_segreduce<Value>(visit, edges, segment_start, frontier_size, reduced, reduce_op(), identity, context);
  return (int)edges;
}

}  // namespace neighborhood
}  // namespace oprtr
}  // namespace gunrock
