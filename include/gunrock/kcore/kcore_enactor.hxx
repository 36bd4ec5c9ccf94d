// gunrock/kcore/kcore_enactor.hxx -- k-core decomposition by peeling, on the filter and advance operators
// (mgx_kcore_enact).  What the reference's kcore_enactor_t::enact does (gunrock/src/kcore/kcore_enactor.hxx:40-86):
//   for k = 1, 2, ...: passes over ALL vertices until one removes nobody --
//       filter<deg_less_than_k>   the vertices with 0 < degree < k leave: core k - 1, degree 0; they are the pass's frontier
//       advance<update_deg, idempotence = false, has_output = false>   every entry of a leaving vertex takes one degree
//                                 from its neighbour (atomicAdd); no output frontier is written, the operator returns 0
//       filter<deg_atleast_k>     how many vertices still have degree >= k (upstream runs it twice per removing pass;
//                                 the functor has no side effect, so once is the same)
//   the first k whose count is 0 ends the run: largest_k_core = k - 1.
// Upstream `selector` never flips and the advance writes nothing, so buffers[0] holds 0 .. n-1 throughout; upstream
// re-uploads that iota from the host for every k (init_frontier, :33-38) -- here it is loaded once, device to device.
// Quirk kept: the count is only refreshed by a pass that removed something, so a k whose first pass removes nobody is
// judged by the previous k's count (n before any): a graph without entries runs to k = n and keeps largest_k_core = -1
// (the CPU validator answers 0 there; oracle/oracle.c restates both).
#pragma once
#include "../advance.hxx"
#include "../enactor.hxx"
#include "../filter.hxx"
#include "../frontier.hxx"
#include "../graph.hxx"
#include "kcore_functor.hxx"
#include "kcore_problem.hxx"

namespace gunrock {
namespace kcore {

struct kcore_enactor_t : enactor_t {
  // what the last enact() did: k values tried, passes, entries expanded, vertices removed
  long long rounds = 0, passes = 0, expanded = 0, removed = 0;
  bool verbose = false;                          // the reference prints "largest k-core: K"

  kcore_enactor_t(standard_context_t& ctx, int num_nodes, int num_edges) : enactor_t(ctx, num_nodes, num_edges) {}
  kcore_enactor_t(const kcore_enactor_t&) = delete;
  kcore_enactor_t& operator=(const kcore_enactor_t&) = delete;

  void enact(std::shared_ptr<kcore_problem_t> problem, standard_context_t& ctx) {
    namespace adv = gunrock::oprtr::advance;
    namespace fl = gunrock::oprtr::filter;
    const int n = problem->gslice->num_nodes;
    frontier_ptr& everyone = indices;            // enactor_t's iota: node capacity is all it needs (upstream: buffers[0])
    frontier_ptr& leaving = filtered_indices;    // at most n vertices leave in a pass (upstream: buffers[1])
    frontier_ptr& unused_output = buffers[0];    // the advance's signature wants one; has_output = false never touches it
    everyone->resize((size_t)n);
    rounds = passes = expanded = removed = 0;
    long long in_the_running = n;
    for (int k = 1; k <= n; ++k) {
      ++rounds;
      for (;;) {
        ++passes;
        const int gone = fl::filter_kernel<kcore_problem_t, deg_less_than_k_functor_t>(problem, everyone, leaving, k, ctx);
        if (gone == 0) break;
        removed += gone;
        adv::advance_forward_kernel<kcore_problem_t, update_deg_functor_t, /*idempotence=*/false, /*has_output=*/false>(
            problem, leaving, unused_output, k, ctx);
        expanded += ctx.mailbox[0];             // the total of the degree scan the advance ran (advance.hxx:43's read-back)
        in_the_running = count_at_least(problem, everyone, k, ctx);
      }
      if (in_the_running == 0) {
        problem->largest_k_core = k - 1;
        if (verbose) std::cout << "largest k-core: " << k - 1 << std::endl;
        break;
      }
    }
  }

 private:
  // deg_atleast_k as a filter writes the survivors somewhere; only their number matters (upstream writes them into
  // the buffer the next pass overwrites), so the compaction's upsweep alone is run: a count, no scatter
  static long long count_at_least(std::shared_ptr<kcore_problem_t>& problem, frontier_ptr& everyone, int k,
                                  standard_context_t& ctx) {
    auto counter = mgx::transform_compact((long long)everyone->size(), ctx);
    const int* const ids = everyone->data()->data();
    kcore_slice_t* const slice = problem->d_data_slice.data();
    return counter.upsweep([=] __device__(long long i) { return deg_atleast_k_functor_t::cond_filter(ids[i], slice, k); });
  }
};

}  // namespace kcore
}  // namespace gunrock
