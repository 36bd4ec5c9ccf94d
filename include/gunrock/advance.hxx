// gunrock/advance.hxx -- the advance operator family on mgx's kernels (HIP, wave64; no moderngpu).
// Drop-in for the reference's gunrock/src/advance.hxx: advance_forward_kernel (:20-67), sparse_to_dense_kernel
// (:69-84), gen_unvisited_kernel (:86-106), advance_backward_kernel (:108-160) -- the same template parameters,
// argument order and return values, and the same functor call contract: cond_advance THEN apply_advance are both
// evaluated for every expanded edge (:57-58).
//
// Extension (not in the reference): advance_filter_fused_kernel does advance + filter in one pass over the edges
// and never materialises the mostly -1 intermediate frontier.
#pragma once
#include <climits>
#include <type_traits>

#include "../mgx/lbs.hpp"
#include "../mgx/scan.hpp"
#include "frontier.hxx"
#include "intrinsics.hxx"

namespace gunrock {
namespace oprtr {
namespace advance {

namespace detail {

// A functor may say that its filter test looks at the slot's value alone (no state another edge of the same advance could still
// change): `static constexpr bool cond_filter_of_slot_value_only = true;` (the repo's bfs_functor_t: value != -1).  The advance
// then evaluates that test per output slot while the value is in a register and leaves the ballots for the filter behind it
// (mgx::standard_context_t::keep).  The reference's unchanged functors say nothing and are filtered the reference's way.
template <typename F, typename = void>
struct filter_of_slot_value_only : std::false_type {};
template <typename F>
struct filter_of_slot_value_only<F, typename std::enable_if<F::cond_filter_of_slot_value_only>::type> : std::true_type {};

// Segment i of an expansion = the adjacency list of ids[i] under `offsets` (CSR rows when pushing, CSC columns when
// pulling).  Writes the exclusive scan of the list lengths into the graph's shared scan buffer (graph.hxx:49-52:
// one expansion at a time per graph) and returns the number of edges; the 8-byte read-back of that total is the
// reference's advance.hxx:43.
inline long long scan_adjacency_sizes(graph_device_t& graph, const int* ids, long long count, size_t capacity,
                                      const int* offsets, standard_context_t& context) {
  graph.ensure_scanned(capacity, context);
  long long edges = 0;
  mgx::transform_scan([=] __device__(long long i) { return offsets[ids[i] + 1] - offsets[ids[i]]; }, count,
                      graph.d_scanned_row_offsets.data(), context, &edges);
  return edges;
}

}  // namespace detail

template <typename Problem, typename Functor, bool idempotence, bool has_output>
int advance_forward_kernel(std::shared_ptr<Problem> problem, std::shared_ptr<frontier_t<int>>& input,
                           std::shared_ptr<frontier_t<int>>& output, int iteration, standard_context_t& context) {
  graph_device_t& graph = *problem->gslice;
  const int* const frontier = input->data()->data();
  const long long frontier_size = (long long)input->size();
  const int* const row_start = graph.d_row_offsets.data();
  const long long edges = detail::scan_adjacency_sizes(graph, frontier, frontier_size, input->capacity(), row_start, context);
  if (edges == 0) {
    if (has_output) output->resize(0);
    return 0;
  }
  if (edges > INT_MAX) throw mgx::mgx_error(MGX_E_FRONTIER_OVERFLOW, "advance: more than 2^31-1 work items");

  int* out = nullptr;
  if (has_output) {
    output->resize((size_t)edges);              // one slot per EDGE, as upstream: losers become -1 for the filter
    out = output->data()->data();
  }
  const int* const neighbours = graph.d_col_indices.data();
  typename Problem::data_slice_t* const data = problem->d_data_slice.data();
  // (Staging (v, row start) per segment in LDS instead of the two gathers per edge below changed nothing.  The time
  //  of this kernel is the functor's: the contract calls apply_advance for EVERY edge, whatever cond_advance said,
  //  and the reference's bfs_functor_t does an atomicCAS on labels[dst] there -- 134 M device-scope atomics per
  //  RMAT-22 traversal at ~25 G/s: 16.5 ms.  Our restatement of the functor reads the label first
  //  (bfs/bfs_functor.hxx): 2.5 ms.)
  context.keep.valid = false;
  if constexpr (has_output && !idempotence && detail::filter_of_slot_value_only<Functor>::value) {
    // the filter's test rides on the expansion: one ballot word per 64 slots into the compaction's bit array (scan.hpp: compact_t
    // lays it out the same way for the same count), nothing else of the filter's upsweep is left but to count them
    mgx::compact_t where(edges, context);
    auto visit_keep = [=] __device__(int slot, int segment, int rank) -> bool {
      const int v = frontier[segment];
      const int edge = row_start[v] + rank;
      const int u = neighbours[edge];
      const bool cond = Functor::cond_advance(v, u, edge, rank, slot, data, iteration);
      const bool applied = Functor::apply_advance(v, u, edge, rank, slot, data, iteration);
      const int value = (cond && applied) ? u : -1;
      out[slot] = value;
      return Functor::cond_filter(value, data, iteration);
    };
    ++context.scratch_epoch;
    mgx::transform_lbs_keep(visit_keep, edges, graph.d_scanned_row_offsets.data(), frontier_size, where.bits, context);
    context.keep.data = out; context.keep.n = edges; context.keep.iteration = iteration;
    context.keep.functor = &mgx::functor_tag_t<Functor>::id; context.keep.epoch = context.scratch_epoch; context.keep.generation = mgx::frontier_generation(); context.keep.valid = true;
    return (int)edges;
  }
  auto visit = [=] __device__(int slot, int segment, int rank) {
    const int v = frontier[segment];
    const int edge = row_start[v] + rank;
    const int u = neighbours[edge];
    const bool cond = Functor::cond_advance(v, u, edge, rank, slot, data, iteration);
    const bool applied = Functor::apply_advance(v, u, edge, rank, slot, data, iteration);
    if (has_output) out[slot] = (idempotence || (cond && applied)) ? u : -1;
  };
  mgx::transform_lbs(visit, edges, graph.d_scanned_row_offsets.data(), frontier_size, context);
  return has_output ? (int)edges : 0;
}

// advance + filter(idx != -1) in one pass: output holds exactly the neighbours for which cond_advance &&
// apply_advance held (order unspecified).  Returns the output length.
template <typename Problem, typename Functor>
int advance_filter_fused_kernel(std::shared_ptr<Problem> problem, std::shared_ptr<frontier_t<int>>& input,
                                std::shared_ptr<frontier_t<int>>& output, int iteration,
                                standard_context_t& context) {
  graph_device_t& graph = *problem->gslice;
  const int* const frontier = input->data()->data();
  const long long frontier_size = (long long)input->size();
  const int* const row_start = graph.d_row_offsets.data();
  const long long edges = detail::scan_adjacency_sizes(graph, frontier, frontier_size, input->capacity(), row_start, context);
  if (edges == 0) {
    output->resize(0);
    return 0;
  }
  const int* const neighbours = graph.d_col_indices.data();
  int* const out = output->data()->data();
  const long long room = (long long)output->capacity();
  typename Problem::data_slice_t* const data = problem->d_data_slice.data();
  // the append cursor lives in the scratch arena, behind the scan's partial sums
  unsigned long long* const cursor =
      (unsigned long long*)((char*)context.scratch + ((size_t)mgx::scan_num_tiles(frontier_size) + 2) * 8);
  ++context.scratch_epoch;
  MGX_HIP(hipMemsetAsync(cursor, 0, sizeof(unsigned long long), context.stream()));
  auto visit_and_keep = [=] __device__(int slot, int segment, int rank) {
    const int v = frontier[segment];
    const int edge = row_start[v] + rank;
    const int u = neighbours[edge];
    const bool cond = Functor::cond_advance(v, u, edge, rank, slot, data, iteration);
    const bool applied = Functor::apply_advance(v, u, edge, rank, slot, data, iteration);
    const bool keep = cond && applied;
    const mgx::u64 keepers = __ballot(keep);          // one cursor atomic per wave that keeps anything
    if (keepers) {
      const int first = __ffsll((long long)keepers) - 1;
      unsigned long long base = 0;
      if (mgx::lane_id() == first) base = atomicAdd(cursor, (unsigned long long)__popcll(keepers));
      base = __shfl(base, first, mgx::WAVE);
      const long long at = (long long)base + mgx::rank_in_mask(keepers);
      if (keep && at < room) out[at] = u;
    }
  };
  mgx::transform_lbs(visit_and_keep, edges, graph.d_scanned_row_offsets.data(), frontier_size, context);
  MGX_HIP(hipMemcpyAsync(context.mailbox, cursor, sizeof(long long), hipMemcpyDeviceToHost, context.stream()));
  context.synchronize();
  const long long kept = context.mailbox[0];
  output->resize((size_t)kept);   // throws on overflow; nothing was written past the capacity
  return (int)kept;
}

// dense[v] = cond_sparse_to_dense(v) for the vertices listed in `sparse` (the pull phase's frontier bitmap: one int
// per vertex, as upstream)
template <typename Problem, typename Functor>
void sparse_to_dense_kernel(std::shared_ptr<Problem> problem, std::shared_ptr<frontier_t<int>>& sparse,
                            std::shared_ptr<frontier_t<int>>& dense, int iteration, standard_context_t& context) {
  const int* const listed = sparse->data()->data();
  int* const flags = dense->data()->data();
  typename Problem::data_slice_t* const data = problem->d_data_slice.data();
  mgx::transform(
      [=] __device__(int i) {
        const int v = listed[i];
        flags[v] = Functor::cond_sparse_to_dense(v, data, iteration) ? 1 : 0;
      },
      (long long)sparse->size(), context);
}

// unvisited <- the members of `indices` for which cond_gen_unvisited holds, in order; returns how many
template <typename Problem, typename Functor>
int gen_unvisited_kernel(std::shared_ptr<Problem> problem, std::shared_ptr<frontier_t<int>>& indices,
                         std::shared_ptr<frontier_t<int>>& unvisited, int iteration, standard_context_t& context) {
  const int* const candidates = indices->data()->data();
  typename Problem::data_slice_t* const data = problem->d_data_slice.data();
  auto compaction = mgx::transform_compact((long long)indices->size(), context);
  const long long kept = compaction.upsweep(
      [=] __device__(long long i) { return Functor::cond_gen_unvisited(candidates[i], data, iteration); });
  unvisited->resize((size_t)kept);
  int* const out = unvisited->data()->data();
  compaction.downsweep([=] __device__(long long to, long long from) { out[to] = candidates[from]; });
  return (int)kept;
}

// One bottom-up step: every in-edge (u -> v) of every vertex v in `unvisited`; the first u found in the frontier
// bitmap for which apply_advance(u, v) succeeds puts v into bitmap_out and retires v's slot.  Returns the number of
// in-edges enumerated.
template <typename Problem, typename Functor>
int advance_backward_kernel(std::shared_ptr<Problem> problem, std::shared_ptr<frontier_t<int>>& unvisited,
                            std::shared_ptr<frontier_t<int>>& bitmap, std::shared_ptr<frontier_t<int>>& bitmap_out,
                            int iteration, standard_context_t& context) {
  graph_device_t& graph = *problem->gslice;
  int* const pending = unvisited->data()->data();
  const long long pending_size = (long long)unvisited->size();
  const int* const column_start = graph.d_col_offsets.data();
  const long long edges =
      detail::scan_adjacency_sizes(graph, pending, pending_size, unvisited->capacity(), column_start, context);
  if (edges == 0) return 0;

  const int* const in_neighbours = graph.d_row_indices.data();
  const int* const in_frontier = bitmap->data()->data();
  int* const next_frontier = bitmap_out->data()->data();
  typename Problem::data_slice_t* const data = problem->d_data_slice.data();
  // The reference re-reads unvisited[segment] for every edge while other lanes set it to -1 (advance.hxx:143,151):
  // a lane that reads the -1 indexes col_offsets[-1].  Here a retired slot ends the lane's work instead.
  auto visit = [=] __device__(int slot, int segment, int rank) {
    const int v = pending[segment];
    if (v < 0) return;                                   // claimed by another in-neighbour during this call
    const int edge = column_start[v] + rank;
    const int u = in_neighbours[edge];
    if (in_frontier[u] && Functor::apply_advance(u, v, edge, rank, slot, data, iteration)) {
      next_frontier[v] = 1;
      pending[segment] = -1;
    }
    (void)Functor::cond_advance(u, v, edge, rank, slot, data, iteration);     // called for its contract, as upstream
  };
  mgx::transform_lbs(visit, edges, graph.d_scanned_row_offsets.data(), pending_size, context);
  return (int)edges;
}

}  // namespace advance
}  // namespace oprtr
}  // namespace gunrock
