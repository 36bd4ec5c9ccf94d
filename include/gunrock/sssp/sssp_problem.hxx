// gunrock/sssp/sssp_problem.hxx -- SSSP state behind the C-ABI (mgx_sssp_*).
// Semantics of the reference's sssp_problem_t (gunrock/src/sssp/sssp_problem.hxx:11-57): distances ("labels")
// are FLT_MAX except 0 at the source, preds -1, the per-iteration dedup stamps -1; the functor also needs the edge
// weights, which stay where the graph keeps them.  The reference's CPU validator (cpu(), :59-88) is restated in
// oracle/oracle.c (orc_sssp_cpu), with the tests.
#pragma once
#include <limits>

#include "../problem.hxx"

namespace gunrock {
namespace sssp {

struct sssp_problem_t : problem_t {
  struct data_slice_t {
    float* d_labels;      // distances
    int* d_preds;
    float* d_weights;     // graph_device_t::d_col_values
    int* d_visited;       // iteration in which the vertex last passed the filter
  };

  int src = 0;
  mem_t<float> d_labels;
  mem_t<int> d_preds, d_visited;
  mem_t<data_slice_t> d_data_slice;
  std::vector<float> labels;               // host copies, filled by extract()
  std::vector<int> preds;

  sssp_problem_t(std::shared_ptr<graph_device_t> graph, size_t source, standard_context_t& ctx) : problem_t(graph) {
    const size_t n = (size_t)graph->num_nodes;
    d_labels = mem_t<float>(n, ctx);
    d_preds = mem_t<int>(n, ctx);
    d_visited = mem_t<int>(n, ctx);
    const data_slice_t slice{d_labels.data(), d_preds.data(), graph->d_col_values.data(), d_visited.data()};
    d_data_slice = to_mem(std::vector<data_slice_t>(1, slice), ctx);
    reset(source, ctx);
  }
  sssp_problem_t(const sssp_problem_t&) = delete;
  sssp_problem_t& operator=(const sssp_problem_t&) = delete;

  // the state a fresh problem has, for another source: one pass over the three arrays
  void reset(size_t source, standard_context_t& ctx) {
    src = (int)source;
    const int s = src;
    float* const dist = d_labels.data();
    int* const pred = d_preds.data();
    int* const stamp = d_visited.data();
    mgx::transform(
        [=] __device__(int v) {
          dist[v] = (v == s) ? 0.0f : std::numeric_limits<float>::max();
          pred[v] = -1;
          stamp[v] = -1;
        },
        gslice->num_nodes, ctx);
  }

  void extract() {
    const size_t n = (size_t)gslice->num_nodes;
    MGX_HIP(mgx::dtoh(labels, d_labels.data(), n));
    MGX_HIP(mgx::dtoh(preds, d_preds.data(), n));
  }
};

}  // namespace sssp
}  // namespace gunrock
