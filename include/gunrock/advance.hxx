// gunrock/advance.hxx -- the advance operator family.
// Drop-in for the reference's gunrock/src/advance.hxx (advance_forward_kernel :20-67,
// sparse_to_dense_kernel :69-84, gen_unvisited_kernel :86-106, advance_backward_kernel
// :108-160): same template parameters, argument order, return values and functor call
// contract -- cond_advance THEN apply_advance are both evaluated for every expanded edge
// (:57-58).  The kernels underneath are mgx's (HIP, wave64), not moderngpu's.
//
// Extension (not in the reference): advance_filter_fused_kernel does advance + filter in one
// pass over the edges and never materialises the mostly -1 intermediate frontier.
#pragma once
#include <climits>

#include "../mgx/lbs.hpp"
#include "../mgx/scan.hpp"
#include "frontier.hxx"
#include "intrinsics.hxx"

namespace gunrock {
namespace oprtr {
namespace advance {

template <typename Problem, typename Functor, bool idempotence, bool has_output>
int advance_forward_kernel(std::shared_ptr<Problem> problem, std::shared_ptr<frontier_t<int>>& input,
                           std::shared_ptr<frontier_t<int>>& output, int iteration, standard_context_t& context) {
  const int* input_data = input->data()->data();
  problem->gslice->ensure_scanned(input->capacity(), context);
  int* scanned_row_offsets = problem->gslice->d_scanned_row_offsets.data();
  const int* row_offsets = problem->gslice->d_row_offsets.data();

  long long front = 0;
  mgx::transform_scan(
      [=] __device__(long long idx) {
        const int v = input_data[idx];
        return row_offsets[v + 1] - row_offsets[v];
      },
      (long long)input->size(), scanned_row_offsets, context, &front);

  if (!front) {
    if (has_output) output->resize(0);
    return 0;
  }
  if (front > INT_MAX) throw mgx::mgx_error(MGX_E_FRONTIER_OVERFLOW, "advance: more than 2^31-1 work items");
  if (has_output) output->resize((size_t)front);

  const int* col_indices = problem->gslice->d_col_indices.data();
  int* output_data = has_output ? output->data()->data() : nullptr;
  typename Problem::data_slice_t* data = problem->d_data_slice.data();
  // (Staging (v, row start) per segment in LDS instead of the two gathers per edge below changed nothing.  The time
  //  of this kernel is the functor's: the contract calls apply_advance for EVERY edge, whatever cond_advance said
  //  (advance.hxx:57-58), and the reference's bfs_functor_t does an atomicCAS on labels[dst] there -- 134 M
  //  device-scope atomics per RMAT-22 traversal at ~25 G/s: 16.5 ms.  Our restatement of the functor reads the label
  //  first (bfs/bfs_functor.hxx): 2.5 ms.)
  auto neighbors_expand = [=] __device__(int idx, int seg, int rank) {
    const int v = input_data[seg];
    const int start_idx = row_offsets[v];
    const int neighbor = col_indices[start_idx + rank];
    const bool cond = Functor::cond_advance(v, neighbor, start_idx + rank, rank, idx, data, iteration);
    const bool apply_return = Functor::apply_advance(v, neighbor, start_idx + rank, rank, idx, data, iteration);
    if (has_output) output_data[idx] = idempotence ? neighbor : ((cond && apply_return) ? neighbor : -1);
  };
  mgx::transform_lbs(neighbors_expand, front, scanned_row_offsets, (long long)input->size(), context);

  if (!has_output) front = 0;
  return (int)front;
}

// advance + filter(idx != -1) in one pass: output holds exactly the neighbours for which
// cond_advance && apply_advance held (order unspecified).  Returns the output length.
template <typename Problem, typename Functor>
int advance_filter_fused_kernel(std::shared_ptr<Problem> problem, std::shared_ptr<frontier_t<int>>& input,
                                std::shared_ptr<frontier_t<int>>& output, int iteration,
                                standard_context_t& context) {
  const int* input_data = input->data()->data();
  problem->gslice->ensure_scanned(input->capacity(), context);
  int* scanned_row_offsets = problem->gslice->d_scanned_row_offsets.data();
  const int* row_offsets = problem->gslice->d_row_offsets.data();
  long long front = 0;
  mgx::transform_scan(
      [=] __device__(long long idx) {
        const int v = input_data[idx];
        return row_offsets[v + 1] - row_offsets[v];
      },
      (long long)input->size(), scanned_row_offsets, context, &front);
  if (!front) {
    output->resize(0);
    return 0;
  }
  const int* col_indices = problem->gslice->d_col_indices.data();
  int* output_data = output->data()->data();
  const long long cap = (long long)output->capacity();
  typename Problem::data_slice_t* data = problem->d_data_slice.data();
  // device-side append cursor lives in the scratch arena, after the scan partials
  unsigned long long* cursor =
      (unsigned long long*)((char*)context.scratch + (((size_t)mgx::scan_num_tiles((long long)input->size()) + 2) * 8));
  MGX_HIP(hipMemsetAsync(cursor, 0, sizeof(unsigned long long), context.stream()));
  auto expand_keep = [=] __device__(int idx, int seg, int rank) {
    const int v = input_data[seg];
    const int start_idx = row_offsets[v];
    const int neighbor = col_indices[start_idx + rank];
    const bool cond = Functor::cond_advance(v, neighbor, start_idx + rank, rank, idx, data, iteration);
    const bool app = Functor::apply_advance(v, neighbor, start_idx + rank, rank, idx, data, iteration);
    const bool keep = cond && app;
    // wave-aggregated append: one atomic per wave that has anything to keep
    const mgx::u64 m = __ballot(keep);
    if (m) {
      const int leader = __ffsll((long long)m) - 1;
      unsigned long long base = 0;
      if (mgx::lane_id() == leader) base = atomicAdd(cursor, (unsigned long long)__popcll(m));
      base = __shfl(base, leader, mgx::WAVE);
      if (keep) {
        const long long dst = (long long)base + mgx::rank_in_mask(m);
        if (dst < cap) output_data[dst] = neighbor;
      }
    }
  };
  mgx::transform_lbs(expand_keep, front, scanned_row_offsets, (long long)input->size(), context);
  MGX_HIP(hipMemcpyAsync(context.mailbox, cursor, sizeof(long long), hipMemcpyDeviceToHost, context.stream()));
  context.synchronize();
  const long long kept = context.mailbox[0];
  output->resize((size_t)kept);   // throws on overflow; nothing was written past capacity
  return (int)kept;
}

template <typename Problem, typename Functor>
void sparse_to_dense_kernel(std::shared_ptr<Problem> problem, std::shared_ptr<frontier_t<int>>& sparse,
                            std::shared_ptr<frontier_t<int>>& dense, int iteration, standard_context_t& context) {
  const int* input_data = sparse->data()->data();
  int* output_data = dense->data()->data();
  typename Problem::data_slice_t* data = problem->d_data_slice.data();
  mgx::transform(
      [=] __device__(int idx) {
        const int item = input_data[idx];
        output_data[item] = Functor::cond_sparse_to_dense(item, data, iteration) ? 1 : 0;
      },
      (long long)sparse->size(), context);
}

template <typename Problem, typename Functor>
int gen_unvisited_kernel(std::shared_ptr<Problem> problem, std::shared_ptr<frontier_t<int>>& indices,
                         std::shared_ptr<frontier_t<int>>& unvisited, int iteration, standard_context_t& context) {
  auto compact = mgx::transform_compact((long long)indices->size(), context);
  const int* input_data = indices->data()->data();
  typename Problem::data_slice_t* data = problem->d_data_slice.data();
  const long long stream_count = compact.upsweep([=] __device__(long long idx) {
    const int item = input_data[idx];
    return Functor::cond_gen_unvisited(item, data, iteration);
  });
  unvisited->resize((size_t)stream_count);
  int* unvisited_data = unvisited->data()->data();
  compact.downsweep(
      [=] __device__(long long dest_idx, long long source_idx) { unvisited_data[dest_idx] = input_data[source_idx]; });
  return (int)stream_count;
}

template <typename Problem, typename Functor>
int advance_backward_kernel(std::shared_ptr<Problem> problem, std::shared_ptr<frontier_t<int>>& unvisited,
                            std::shared_ptr<frontier_t<int>>& bitmap, std::shared_ptr<frontier_t<int>>& bitmap_out,
                            int iteration, standard_context_t& context) {
  int* unvisited_data = unvisited->data()->data();
  problem->gslice->ensure_scanned(unvisited->capacity(), context);
  int* scanned_row_offsets = problem->gslice->d_scanned_row_offsets.data();
  const int* col_offsets = problem->gslice->d_col_offsets.data();

  long long front = 0;
  mgx::transform_scan(
      [=] __device__(long long idx) {
        const int v = unvisited_data[idx];
        return col_offsets[v + 1] - col_offsets[v];
      },
      (long long)unvisited->size(), scanned_row_offsets, context, &front);
  if (!front) return 0;

  const int* row_indices = problem->gslice->d_row_indices.data();
  typename Problem::data_slice_t* data = problem->d_data_slice.data();
  const int* bitmap_data = bitmap->data()->data();
  int* bitmap_out_data = bitmap_out->data()->data();
  // The reference re-reads unvisited_data[seg] inside the expansion while other lanes set it
  // to -1 (advance.hxx:143,151) -- a read of -1 there indexes col_offsets[-1].  The vertex is
  // recovered here from a second, read-only view: v never changes, only its slot is retired.
  auto neighbors_expand = [=] __device__(int idx, int seg, int rank) {
    int v = unvisited_data[seg];
    if (v < 0) return;   // already claimed by another in-neighbour this call
    const int start_idx = col_offsets[v];
    const int neighbor = row_indices[start_idx + rank];
    if (bitmap_data[neighbor] && Functor::apply_advance(neighbor, v, start_idx + rank, rank, idx, data, iteration)) {
      bitmap_out_data[v] = 1;
      unvisited_data[seg] = -1;
    }
    if (!Functor::cond_advance(neighbor, v, start_idx + rank, rank, idx, data, iteration)) return;
  };
  mgx::transform_lbs(neighbors_expand, front, scanned_row_offsets, (long long)unvisited->size(), context);
  return (int)front;
}

}  // namespace advance
}  // namespace oprtr
}  // namespace gunrock
