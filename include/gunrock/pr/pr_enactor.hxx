// gunrock/pr/pr_enactor.hxx -- PageRank-style loop on the neighbourhood-reduce operator (mgx_pr_enact).
// What the reference's pr_enactor_t::enact does (gunrock/src/pr/pr_enactor.hxx:41-79): the frontier starts as all
// vertices; an iteration reduces the neighbours' ranks into d_reduced_ranks -- by frontier POSITION
// (neighborhood.hxx:58) -- and then filters the frontier with pr_functor_t::cond_filter, which rewrites the ranks
// reading d_reduced_ranks by VERTEX id: only iteration 0, where position == id, is a true PageRank step.  That
// upstream quirk is kept; the oracle restates it the same way.
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
  std::vector<long long> frontier_lengths;      // what the filter kept, per iteration
  bool verbose = false;                          // the reference prints a line per iteration

  pr_enactor_t(standard_context_t& ctx, int num_nodes, int num_edges)
      : enactor_t(ctx, num_nodes, num_edges), unused_output(std::make_shared<frontier_t<int>>(ctx, (size_t)num_nodes)) {}
  pr_enactor_t(const pr_enactor_t&) = delete;
  pr_enactor_t& operator=(const pr_enactor_t&) = delete;

  void enact(std::shared_ptr<pr_problem_t> problem, standard_context_t& ctx) {
    namespace nb = gunrock::oprtr::neighborhood;
    namespace fl = gunrock::oprtr::filter;
    const int n = problem->gslice->num_nodes;
    (void)buffers[0]->load(*indices->data());    // 0, 1, ..., n-1 (enactor_t keeps the iota)
    buffers[0]->resize(n);
    frontier_lengths.clear();
    int in = 0;
    for (int it = 0, kept = n; kept > 0 && it < problem->max_iter; ++it, in ^= 1) {
      nb::neighborhood_kernel<pr_problem_t, pr_functor_t, float, mgx::plus_t<float>, false, false>(
          problem, buffers[in], unused_output, problem->d_reduced_ranks.data(), 0.0f, it, ctx);
      kept = fl::filter_kernel<pr_problem_t, pr_functor_t>(problem, buffers[in], buffers[in ^ 1], it, ctx);
      frontier_lengths.push_back(kept);
      if (verbose) std::cout << "finished iteration:" << it << " output length: " << kept << std::endl;
    }
  }

 private:
  std::shared_ptr<frontier_t<int>> unused_output;   // the operator's signature wants an output frontier
};

}  // namespace pr
}  // namespace gunrock
