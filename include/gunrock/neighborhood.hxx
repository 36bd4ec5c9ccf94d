// gunrock/neighborhood.hxx -- segmented neighbourhood reduce.
// Drop-in for the reference's gunrock/src/neighborhood.hxx:12-70: same six template
// parameters (push is not deducible, callers must pass it -- SURVEY F11), same call contract:
// per edge cond_advance, apply_advance, optional output write, get_value_to_reduce(neighbor);
// reduced[] is indexed by FRONTIER POSITION (:58), identity for empty segments; returns the
// number of edges visited.
#pragma once

#include <type_traits>

#include "../mgx/lbs.hpp"
#include "../mgx/env.hpp"
#include "../mgx/nreduce.hpp"
#include "../mgx/scan.hpp"
#include "frontier.hxx"
#include "intrinsics.hxx"

namespace gunrock {
namespace oprtr {
namespace neighborhood {

// A functor may declare `static constexpr bool mgx_pure_gather = true`: its cond_advance / apply_advance are trivially
// true and get_value_to_reduce(v) is a pure read.  Only then may the operator skip the two calls and take a vertex's value
// once instead of once per edge (the full-frontier path below); the reference's own functors declare nothing and always
// get the per-edge contract of neighborhood.hxx:39-57.
template <typename F, typename = void>
struct is_pure_gather : std::false_type {};
template <typename F>
struct is_pure_gather<F, typename std::enable_if<F::mgx_pure_gather>::type> : std::true_type {};

// MGX_NR_SUBSET=0: frontiers that are not 0 .. n - 1 take the general kernel, as before round 6
inline bool nr_subset_enabled() {
  static const bool on = [] { const char* e = mgx::env("MGX_NR_SUBSET"); return !e || std::atoi(e) != 0; }();
  return on;
}

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

  // Full frontier (0, 1, ..., n - 1: PR's first iteration) on a graph that carries the hub-first layout with unit blocks and
  // degree classes: mgx/nreduce.hpp.  Whether the frontier IS the iota is checked on the device (one pass over it);
  // it then holds every edge of the graph: no degree scan, no search.
  typename Problem::data_slice_t* const data = problem->d_data_slice.data();
  // (round 6) ... or a large SUBSET of the vertices in ascending order -- what PR's filter leaves for every iteration but the first
  // (pr_enactor.hxx:53-66): the same kernels compute every row and keep the frontier's, results by frontier position.  A frontier of
  // fewer than n / 8 vertices takes the general kernel, whose cost follows the frontier's edges (the fast path costs what a full
  // frontier costs).
  const bool full = frontier_size == (long long)graph.num_nodes;
  const bool subset = !full && frontier_size < (long long)graph.num_nodes && frontier_size * 8 >= (long long)graph.num_nodes && nr_subset_enabled();
  if constexpr (sizeof(Value) == 4)     // (the kernel keeps 40 000 4-byte values in the 160 KB of LDS: wider values take the general path)
  if (!has_output && is_pure_gather<Functor>::value && (full || subset) && frontier_size > 0 &&
      graph.has_layout && (push || graph.csc_is_csr) && graph.ub_units > 0 && graph.ub_min_degree == graph.vs_long_min && graph.vs_long_min >= 17 && graph.vs_long_min <= 64 &&
      graph.d_ub_cnt.size() && graph.d_ub_first.size() && graph.vs_dummy != 0 &&
      context.scratch_bytes >= mgx::nr_scratch_bytes(graph.num_nodes, graph.ub_units_pad, sizeof(Value))) {
    // the check (inside the first kernel) and the work go out back to back: the kernels behind it look at its verdict themselves (a device
    // word that holds the epoch of the last call whose frontier was NOT the iota); one host wait, behind everything
    context.mailbox[8] = 1;
    context.mailbox[9] = (long long)context.nr_edges_base;
    const unsigned epoch = context.next_nr_epoch();
    if (graph.num_edges > 0) {
      mgx::nr_layout_t L;
      L.new_of_old = graph.d_new_of_old.data();
      L.row_offsets = (const mgx::u32*)graph.d_layout_row_offsets.data();
      L.col_indices = graph.d_layout_col_indices.data();
      L.old_of_new = graph.d_old_of_new.data();
      L.ub_col = graph.d_ub_col.size() ? graph.d_ub_col.data() : nullptr;      // (round 6: gone when the layout carries the 24-bit copy)
      L.ub_col24 = graph.d_ub_col24.size() ? graph.d_ub_col24.data() : nullptr;     // (the 24-bit copy whenever the layout has one: 0.492 -> 0.477 ms in round 4)
      L.ub_cnt = graph.d_ub_cnt.data();
      L.ub_first = graph.d_ub_first.data();
      L.ub_units = (mgx::u32)graph.ub_units; L.ub_units_pad = (mgx::u32)graph.ub_units_pad;
      for (int i = 0; i < 4; ++i) L.vs_v[i] = graph.vs_v[i];
      L.vs_dummy = graph.vs_dummy;
      L.big_rows = graph.nr_big_rows;
      L.n = graph.num_nodes;
      {
        static const unsigned parts = [] { const char* e = mgx::env("MGX_NR_PARTS"); return e ? (unsigned)std::atoi(e) & 3u : 3u; }();
        L.parts = parts ? parts : 3u;
      }
      {
        // the long rows by slice of their destinations (MGX_NR_SLICED=0: the unit blocks, as before round 5), when the graph carries
        // them and the scratch arena holds a partial per mini-unit
        static const bool sliced = [] { const char* e = mgx::env("MGX_NR_SLICED"); return !e || std::atoi(e) != 0; }();
        if (sliced && graph.nrs_units > 0 && graph.nrs_slices > 0 && graph.nrs_rows == graph.vs_v[0] && graph.d_nrs_mu.size() && graph.d_nrs_off.size() &&
            context.scratch_bytes >= mgx::nr_scratch_bytes(graph.num_nodes, graph.nrs_units, sizeof(Value))) {
          L.nrs_mu = (const uint4*)graph.d_nrs_mu.data();
          L.nrs_off = graph.d_nrs_off.data();
          for (int i = 0; i < mgx::NRS_MAX_SLICES + 2; ++i) L.nrs_first[i] = graph.nrs_first[i];
          L.nrs_slices = graph.nrs_slices; L.nrs_rows = graph.nrs_rows; for (int i = 0; i < 3; ++i) L.nrs_tier[i] = graph.nrs_tier[i];
        }
      }
      if (subset && graph.d_nr_pos.size() < (size_t)graph.num_nodes) {       // (first subset call on this graph: the one allocation of this path)
        context.synchronize();
        graph.d_nr_pos = mem_t<unsigned long long>((size_t)graph.num_nodes, context);
        MGX_HIP(hipMemsetAsync(graph.d_nr_pos.data(), 0, (size_t)graph.num_nodes * sizeof(unsigned long long), context.stream()));
      }
      const long long seq = mgx::nr_full_frontier<Value>(L, [=] __device__(int v) -> Value { return Functor::get_value_to_reduce(v, data, iteration); }, reduced,
                                   identity, reduce_op(), context, frontier, context.mailbox + 8, context.nr_flag(), epoch,
                                   full ? -1 : frontier_size, offsets, subset ? (mgx::u64*)graph.d_nr_pos.data() : nullptr);
      if (seq) context.mailbox_wait(seq); else context.synchronize();
      // (a subset call added its frontier's degrees to the context's counter whatever its verdict: the base follows)
      const unsigned long long before = context.nr_edges_base;
      if (subset) context.nr_edges_base = (unsigned long long)context.mailbox[9];
      if (context.mailbox[8] == 1) return full ? (int)graph.num_edges : (int)(context.nr_edges_base - before);
    }
  }

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
