// gunrock/sssp/sssp_enactor.hxx -- SSSP superstep loop.
// enact(): the reference's loop (gunrock/src/sssp/sssp_enactor.hxx:40-72): advance (relax
// with atomicMin) then filter (per-iteration stamp dedup) until the filter returns nothing.
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
  int iterations = 0;
  long long relaxations = 0;       // sum of advance fronts (edge relaxations attempted)
  long long frontier_total = 0;    // sum of input frontier lengths

  sssp_enactor_t(standard_context_t& context, int num_nodes, int num_edges, float queue_sizing)
      : enactor_t(context, num_nodes, num_edges, queue_sizing) {}

  sssp_enactor_t(const sssp_enactor_t& rhs) = delete;
  sssp_enactor_t& operator=(const sssp_enactor_t& rhs) = delete;

  void init_frontier(std::shared_ptr<sssp_problem_t> sssp_problem) {
    std::vector<int> node_idx(1, sssp_problem->src);
    (void)buffers[0]->load(node_idx);
  }

  void enact(std::shared_ptr<sssp_problem_t> sssp_problem, standard_context_t& context) {
    using namespace gunrock::oprtr::advance;
    using namespace gunrock::oprtr::filter;
    init_frontier(sssp_problem);
    int frontier_length = 1;
    int selector = 0;
    int iteration;
    relaxations = frontier_total = 0;
    for (iteration = 0;; ++iteration) {
      frontier_total += (long long)buffers[selector]->size();
      frontier_length = advance_forward_kernel<sssp_problem_t, sssp_functor_t, false, true>(
          sssp_problem, buffers[selector], buffers[selector ^ 1], iteration, context);
      relaxations += frontier_length;
      selector ^= 1;
      frontier_length = filter_kernel<sssp_problem_t, sssp_functor_t>(sssp_problem, buffers[selector],
                                                                     buffers[selector ^ 1], iteration, context);
      if (!frontier_length) break;
      selector ^= 1;
    }
    iterations = iteration + 1;
  }
};

}  // namespace sssp
}  // namespace gunrock
