// gunrock/pr/pr_enactor.hxx -- PageRank-style loop on the neighbourhood-reduce operator.
// enact(): the reference's loop (gunrock/src/pr/pr_enactor.hxx:41-79): frontier = all
// vertices; per iteration neighborhood_kernel<..., float, plus_t<float>, false, false> into
// d_reduced_ranks (indexed by frontier POSITION, neighborhood.hxx:58) then filter_kernel whose
// cond_filter rewrites the ranks (reading d_reduced_ranks by VERTEX id -- the upstream quirk
// that makes only iteration 0 a true PageRank step; kept).
#pragma once
#include "../enactor.hxx"
#include "../filter.hxx"
#include "../frontier.hxx"
#include "../graph.hxx"
#include "../neighborhood.hxx"
#include "pr_functor.hxx"
#include "pr_problem.hxx"

namespace gunrock {
namespace pr {

struct pr_enactor_t : enactor_t {
  std::vector<long long> frontier_lengths;   // filter output length per iteration
  bool verbose = false;
  std::shared_ptr<frontier_t<int>> dummy_frontier;

  pr_enactor_t(standard_context_t& context, int num_nodes, int num_edges)
      : enactor_t(context, num_nodes, num_edges),
        dummy_frontier(std::make_shared<frontier_t<int>>(context, (size_t)num_nodes)) {}

  pr_enactor_t(const pr_enactor_t& rhs) = delete;
  pr_enactor_t& operator=(const pr_enactor_t& rhs) = delete;

  void init_frontier(std::shared_ptr<pr_problem_t> pr_problem) {
    (void)buffers[0]->load(*indices->data());   // iota
    buffers[0]->resize(pr_problem->gslice->num_nodes);
  }

  void enact(std::shared_ptr<pr_problem_t> pr_problem, standard_context_t& context) {
    using namespace gunrock::oprtr::filter;
    using namespace gunrock::oprtr::neighborhood;
    init_frontier(pr_problem);
    int frontier_length = pr_problem->gslice->num_nodes;
    int selector = 0;
    int iteration = 0;
    float* reduced_ranks = pr_problem->d_reduced_ranks.data();
    frontier_lengths.clear();
    while (frontier_length > 0 && iteration < pr_problem->max_iter) {
      neighborhood_kernel<pr_problem_t, pr_functor_t, float, mgx::plus_t<float>, false, false>(
          pr_problem, buffers[selector], dummy_frontier, reduced_ranks, 0.0f, iteration, context);
      frontier_length = filter_kernel<pr_problem_t, pr_functor_t>(pr_problem, buffers[selector],
                                                                 buffers[selector ^ 1], iteration, context);
      if (verbose)
        std::cout << "finished iteration:" << iteration << " output length: " << frontier_length << std::endl;
      frontier_lengths.push_back(frontier_length);
      ++iteration;
      selector ^= 1;
    }
  }
};

}  // namespace pr
}  // namespace gunrock
