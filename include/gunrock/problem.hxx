// gunrock/problem.hxx -- problem_t, the base of every problem: which graph it runs on, plus GetDegrees.
// Drop-in for the reference's gunrock/src/problem.hxx:7-31 (member `gslice`, the two constructors, GetDegrees).
#pragma once

#include "graph.hxx"

namespace gunrock {

struct problem_t {
  std::shared_ptr<graph_device_t> gslice;          // the device graph the problem reads

  problem_t() : gslice(new graph_device_t()) {}
  explicit problem_t(std::shared_ptr<graph_device_t> graph) : gslice(std::move(graph)) {}
  problem_t(const problem_t&) = delete;
  problem_t& operator=(const problem_t&) = delete;

  // out[v] = out-degree of v.  The reference's takes mem_t<float> only (problem.hxx:23-30), yet its k-core problem
  // calls it with mem_t<int> (kcore_problem.hxx:44) -- any arithmetic element type is accepted here.
  template <typename T>
  void GetDegrees(mem_t<T>& out, standard_context_t& context) {
    const int* const row_offsets = gslice->d_row_offsets.data();
    T* const dst = out.data();
    transform([=] __device__(int v) { dst[v] = (T)(row_offsets[v + 1] - row_offsets[v]); }, gslice->num_nodes, context);
  }
};

}  // namespace gunrock
