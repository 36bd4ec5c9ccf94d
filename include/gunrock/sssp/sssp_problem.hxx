// gunrock/sssp/sssp_problem.hxx -- SSSP problem state.
// Mirrors the reference's sssp_problem_t (gunrock/src/sssp/sssp_problem.hxx:11-57):
// d_labels (float distances, FLT_MAX / 0 at src), d_preds (-1), d_visited (-1 stamps),
// data_slice_t {d_labels, d_preds, d_weights, d_visited}; extract() copies labels and preds.
// The reference's CPU validator (cpu(), :59-88) is restated in oracle/oracle.c (orc_sssp_cpu).
#pragma once
#include <limits>

#include "../problem.hxx"

namespace gunrock {
namespace sssp {

struct sssp_problem_t : problem_t {
  mem_t<float> d_labels;
  mem_t<int> d_preds;
  mem_t<int> d_visited;
  std::vector<float> labels;
  std::vector<int> preds;
  int src;

  struct data_slice_t {
    float* d_labels;
    int* d_preds;
    float* d_weights;
    int* d_visited;
    void init(mem_t<float>& _labels, mem_t<int>& _preds, mem_t<float>& _weights, mem_t<int>& _visited) {
      d_labels = _labels.data();
      d_preds = _preds.data();
      d_weights = _weights.data();
      d_visited = _visited.data();
    }
  };

  mem_t<data_slice_t> d_data_slice;
  std::vector<data_slice_t> data_slice;

  sssp_problem_t() {}
  sssp_problem_t(const sssp_problem_t& rhs) = delete;
  sssp_problem_t& operator=(const sssp_problem_t& rhs) = delete;

  sssp_problem_t(std::shared_ptr<graph_device_t> rhs, size_t src, standard_context_t& context)
      : problem_t(rhs), src((int)src), data_slice(std::vector<data_slice_t>(1)) {
    d_labels = mem_t<float>(rhs->num_nodes, context);
    d_preds = mem_t<int>(rhs->num_nodes, context);
    d_visited = mem_t<int>(rhs->num_nodes, context);
    data_slice[0].init(d_labels, d_preds, gslice->d_col_values, d_visited);
    d_data_slice = to_mem(data_slice, context);
    reset(src, context);
  }

  void reset(size_t new_src, standard_context_t& context) {
    src = (int)new_src;
    const int n = gslice->num_nodes;
    float* lab = d_labels.data();
    int* pr = d_preds.data();
    int* vis = d_visited.data();
    const int s = src;
    mgx::transform(
        [=] __device__(int i) {
          lab[i] = (i == s) ? 0.0f : std::numeric_limits<float>::max();
          pr[i] = -1;
          vis[i] = -1;
        },
        n, context);
  }

  void extract() {
    MGX_HIP(mgx::dtoh(labels, d_labels.data(), gslice->num_nodes));
    MGX_HIP(mgx::dtoh(preds, d_preds.data(), gslice->num_nodes));
  }
};

}  // namespace sssp
}  // namespace gunrock
