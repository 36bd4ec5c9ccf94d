// gunrock/kcore/kcore_problem.hxx -- state of the k-core decomposition the C-ABI runs (mgx_kcore_*).
// Semantics of the reference's kcore_problem_t (gunrock/src/kcore/kcore_problem.hxx:12-52): per vertex a core number
// (starts at 0: a vertex without entries keeps it) and a working degree that starts as the CSR row length -- every
// entry counts, parallel edges and self-loops included (problem.hxx:23-30) -- and that the peeling drives to 0 and
// below; `largest_k_core` is -1 until an enact() has found it.  The device functors see the two arrays through a
// one-element data_slice_t in device memory.  The reference keeps its CPU validator in the same header (cpu(), :54-105);
// here the validator lives with the tests (oracle/oracle.c: orc_kcore_cpu), not in the product.
#pragma once
#include "../problem.hxx"

namespace gunrock {
namespace kcore {

struct kcore_problem_t : problem_t {
  struct data_slice_t {        // what the three functors dereference on the device, both indexed by vertex id
    int* d_num_cores;
    int* d_degrees;
  };

  int largest_k_core = -1;
  mem_t<int> d_num_cores, d_degrees;
  mem_t<data_slice_t> d_data_slice;
  std::vector<int> num_cores;              // host copy, filled by extract()

  kcore_problem_t(std::shared_ptr<graph_device_t> graph, standard_context_t& ctx) : problem_t(graph) {
    const size_t n = (size_t)graph->num_nodes;
    d_num_cores = mem_t<int>(n, ctx);
    d_degrees = mem_t<int>(n, ctx);
    d_data_slice = to_mem(std::vector<data_slice_t>(1, data_slice_t{d_num_cores.data(), d_degrees.data()}), ctx);
    reset(ctx);
  }
  kcore_problem_t(const kcore_problem_t&) = delete;
  kcore_problem_t& operator=(const kcore_problem_t&) = delete;

  // the state a fresh problem has (asynchronous on the context's stream): a decomposition consumes the degrees
  void reset(standard_context_t& ctx) {
    largest_k_core = -1;
    MGX_HIP(hipMemsetAsync(d_num_cores.data(), 0, (size_t)gslice->num_nodes * sizeof(int), ctx.stream()));
    GetDegrees(d_degrees, ctx);
  }

  void extract() { MGX_HIP(mgx::dtoh(num_cores, d_num_cores.data(), (size_t)gslice->num_nodes)); }
};

}  // namespace kcore
}  // namespace gunrock
