// gunrock/bfs/bfs_problem.hxx -- BFS problem state.
// Mirrors the reference's bfs_problem_t (gunrock/src/bfs/bfs_problem.hxx:9-50): d_labels,
// d_preds, host labels/preds, src, data_slice_t {d_labels, d_preds} uploaded as a one-element
// device array, extract().  Labels start at -1 with labels[src] = 0 (:38-40); d_preds is
// initialised to -1 and, as in the reference, no BFS functor ever writes it.
// The reference also keeps its CPU validation routine here (cpu(), :52-72); in this build the
// validator lives with the tests (oracle/oracle.c orc_bfs_cpu), not in the product header.
#pragma once

#include "../problem.hxx"

namespace gunrock {
namespace bfs {

struct bfs_problem_t : problem_t {
  mem_t<int> d_labels;
  mem_t<int> d_preds;
  std::vector<int> labels;
  std::vector<int> preds;
  int src;

  struct data_slice_t {
    int* d_labels;
    int* d_preds;
    void init(mem_t<int>& _labels, mem_t<int>& _preds) {
      d_labels = _labels.data();
      d_preds = _preds.data();
    }
  };

  mem_t<data_slice_t> d_data_slice;
  std::vector<data_slice_t> data_slice;

  bfs_problem_t() {}
  bfs_problem_t(const bfs_problem_t& rhs) = delete;
  bfs_problem_t& operator=(const bfs_problem_t& rhs) = delete;

  bfs_problem_t(std::shared_ptr<graph_device_t> rhs, size_t src, standard_context_t& context)
      : problem_t(rhs), src((int)src), data_slice(std::vector<data_slice_t>(1)) {
    d_labels = mem_t<int>(rhs->num_nodes, context);
    d_preds = mem_t<int>(rhs->num_nodes, context);
    data_slice[0].init(d_labels, d_preds);
    d_data_slice = to_mem(data_slice, context);
    reset(src, context);
  }

  // back to the state the constructor leaves: everything -1, labels[src] = 0
  void reset(size_t new_src, standard_context_t& context) {
    src = (int)new_src;
    const size_t n = gslice->num_nodes;
    MGX_HIP(hipMemsetAsync(d_labels.data(), 0xFF, n * sizeof(int), context.stream()));
    MGX_HIP(hipMemsetAsync(d_preds.data(), 0xFF, n * sizeof(int), context.stream()));
    MGX_HIP(hipMemsetAsync(d_labels.data() + src, 0, sizeof(int), context.stream()));
  }

  void extract() {
    MGX_HIP(mgx::dtoh(labels, d_labels.data(), gslice->num_nodes));
    MGX_HIP(mgx::dtoh(preds, d_preds.data(), gslice->num_nodes));
  }
};

}  // namespace bfs
}  // namespace gunrock
