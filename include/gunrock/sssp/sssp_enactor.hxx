// gunrock/sssp/sssp_enactor.hxx -- SSSP through the operators, one advance and one filter call per superstep
// (mgx_sssp_enact; the fused device loop is mgx/sssp_fused.hpp).
// What the reference's sssp_enactor_t::enact does (gunrock/src/sssp/sssp_enactor.hxx:40-72): relax the frontier's
// out-edges (advance with sssp_functor_t: atomicMin on the distance, one output slot per edge), drop the losers and
// the duplicates (filter: -1 entries and the per-iteration stamp), repeat until the filter keeps nothing.
#pragma once
#include "../advance.hxx"
#include "../enactor.hxx"
#include "../filter.hxx"
#include "../frontier.hxx"
#include "../graph.hxx"
#include "sssp_functor.hxx"
#include "sssp_problem.hxx"

namespace gunrock {
namespace sssp {

struct sssp_enactor_t : enactor_t {
  // statistics of the last enact()
  int iterations = 0;              // supersteps run
  long long relaxations = 0;       // edges handed to the functor (sum of the advance outputs)
  long long frontier_total = 0;    // vertices expanded (sum of the advance inputs)

  sssp_enactor_t(standard_context_t& ctx, int num_nodes, int num_edges, float queue_sizing)
      : enactor_t(ctx, num_nodes, num_edges, queue_sizing) {}
  sssp_enactor_t(const sssp_enactor_t&) = delete;
  sssp_enactor_t& operator=(const sssp_enactor_t&) = delete;

  void enact(std::shared_ptr<sssp_problem_t> problem, standard_context_t& ctx) {
    namespace adv = gunrock::oprtr::advance;
    namespace fl = gunrock::oprtr::filter;
    (void)buffers[0]->load(std::vector<int>(1, problem->src));
    iterations = 0;
    relaxations = frontier_total = 0;
    for (int cur = 0;; ++iterations) {                 // buffers[cur] holds the frontier
      frontier_total += (long long)buffers[cur]->size();
      relaxations += adv::advance_forward_kernel<sssp_problem_t, sssp_functor_t, false, true>(
          problem, buffers[cur], buffers[cur ^ 1], iterations, ctx);
      const int kept = fl::filter_kernel<sssp_problem_t, sssp_functor_t>(problem, buffers[cur ^ 1], buffers[cur],
                                                                        iterations, ctx);
      if (kept == 0) { ++iterations; break; }          // (the superstep that found nothing counts, as upstream)
    }
  }
};

}  // namespace sssp
}  // namespace gunrock
