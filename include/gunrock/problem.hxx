// gunrock/problem.hxx -- problem_t base: the graph slice + GetDegrees.
// Drop-in for the reference's gunrock/src/problem.hxx:7-31.
#pragma once

#include "graph.hxx"

namespace gunrock {

struct problem_t {
  std::shared_ptr<graph_device_t> gslice;

  problem_t() : gslice(std::make_shared<graph_device_t>()) {}
  problem_t(const problem_t& rhs) = delete;
  problem_t& operator=(const problem_t& rhs) = delete;
  problem_t(std::shared_ptr<graph_device_t> rhs) { gslice = rhs; }

  // degrees[v] = row_offsets[v+1] - row_offsets[v]   (problem.hxx:23-30: float only; kcore_problem.hxx:44 calls it
  // with mem_t<int>, which does not compile against the reference's own header -- accepted here)
  template <typename T>
  void GetDegrees(mem_t<T>& _degrees, standard_context_t& context) {
    T* degrees = _degrees.data();
    const int* offsets = gslice->d_row_offsets.data();
    transform([=] __device__(int idx) { degrees[idx] = (T)(offsets[idx + 1] - offsets[idx]); },
              gslice->num_nodes, context);
  }
};

}  // namespace gunrock
