// gunrock/bfs/bfs_problem.hxx -- BFS state behind the C-ABI (mgx_bfs_*) and the fused enactor.
// Semantics of the reference's bfs_problem_t (gunrock/src/bfs/bfs_problem.hxx:9-50): labels are -1 except
// labels[src] = 0 (:38-40); preds are -1 and, as upstream, no BFS functor ever writes them; the device
// functor reaches both arrays through a one-element data_slice_t in device memory.  The reference keeps its CPU
// validation routine in the same header (cpu(), :52-72); here the validator lives with the tests
// (oracle/oracle.c: orc_bfs_cpu), not in the product.
#pragma once
#include "../problem.hxx"

namespace gunrock {
namespace bfs {

struct bfs_problem_t : problem_t {
  struct data_slice_t {
    int* d_labels;
    int* d_preds;
  };

  int src = 0;
  mem_t<int> d_labels, d_preds;
  mem_t<data_slice_t> d_data_slice;
  std::vector<int> labels, preds;          // host copies, filled by extract()

  bfs_problem_t(std::shared_ptr<graph_device_t> graph, size_t source, standard_context_t& ctx) : problem_t(graph) {
    const size_t n = (size_t)graph->num_nodes;
    d_labels = mem_t<int>(n, ctx);
    d_preds = mem_t<int>(n, ctx);
    d_data_slice = to_mem(std::vector<data_slice_t>(1, data_slice_t{d_labels.data(), d_preds.data()}), ctx);
    reset(source, ctx);
  }
  bfs_problem_t(const bfs_problem_t&) = delete;
  bfs_problem_t& operator=(const bfs_problem_t&) = delete;

  // the state a fresh problem has, for another source (asynchronous on the context's stream)
  void reset(size_t source, standard_context_t& ctx) {
    src = (int)source;
    const size_t bytes = (size_t)gslice->num_nodes * sizeof(int);
    hipStream_t s = ctx.stream();
    MGX_HIP(hipMemsetAsync(d_preds.data(), 0xFF, bytes, s));
    MGX_HIP(hipMemsetAsync(d_labels.data(), 0xFF, bytes, s));
    MGX_HIP(hipMemsetAsync(d_labels.data() + src, 0, sizeof(int), s));
  }

  void extract() {
    const size_t n = (size_t)gslice->num_nodes;
    MGX_HIP(mgx::dtoh(labels, d_labels.data(), n));
    MGX_HIP(mgx::dtoh(preds, d_preds.data(), n));
  }
};

}  // namespace bfs
}  // namespace gunrock
