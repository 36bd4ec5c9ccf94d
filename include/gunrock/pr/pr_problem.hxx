// gunrock/pr/pr_problem.hxx -- state of the PageRank-style iteration the C-ABI runs (mgx_pr_*).
// Semantics of the reference's pr_problem_t (gunrock/src/pr/pr_problem.hxx:8-49): every rank starts at the
// teleport share 0.15, the reduced ranks at 0, degrees are kept as floats, `max_iter` bounds the loop, and the
// device functor sees the three arrays through a one-element data_slice_t in device memory.
#pragma once
#include "../problem.hxx"

namespace gunrock {
namespace pr {

struct pr_problem_t : problem_t {
  struct data_slice_t {        // what pr_functor_t dereferences on the device, all indexed by vertex id
    float* d_current_ranks;
    float* d_reduced_ranks;
    float* d_degrees;
  };

  int max_iter = 0;
  mem_t<float> d_current_ranks, d_reduced_ranks, d_degrees;
  mem_t<data_slice_t> d_data_slice;

  pr_problem_t(std::shared_ptr<graph_device_t> graph, int iterations, standard_context_t& ctx)
      : problem_t(graph), max_iter(iterations) {
    const size_t n = (size_t)graph->num_nodes;
    d_current_ranks = mgx::fill(0.15f, n, ctx);
    d_reduced_ranks = mgx::fill(0.0f, n, ctx);
    d_degrees = mem_t<float>(n, ctx);
    GetDegrees(d_degrees, ctx);                       // float(row_offsets[v + 1] - row_offsets[v])
    const std::vector<data_slice_t> host(1, data_slice_t{d_current_ranks.data(), d_reduced_ranks.data(), d_degrees.data()});
    d_data_slice = to_mem(host, ctx);
  }
  pr_problem_t(const pr_problem_t&) = delete;
  pr_problem_t& operator=(const pr_problem_t&) = delete;
};

}  // namespace pr
}  // namespace gunrock
