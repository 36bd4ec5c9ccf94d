// gunrock/enactor.hxx -- enactor_t: two edge-capacity ping-pong frontiers and two
// node-capacity index frontiers pre-filled with iota.
// Drop-in for the reference's gunrock/src/enactor.hxx:11-46.
#pragma once

#include "frontier.hxx"
#include "graph.hxx"
#include "problem.hxx"

namespace gunrock {

struct enactor_t {
  std::vector<std::shared_ptr<frontier_t<int>>> buffers;
  std::shared_ptr<frontier_t<int>> indices;
  std::shared_ptr<frontier_t<int>> filtered_indices;
  std::vector<std::shared_ptr<frontier_t<int>>> unvisited;

  enactor_t(standard_context_t& context, int num_nodes, int num_edges, float queue_sizing = 1.0f) {
    init(context, num_nodes, num_edges, queue_sizing);
  }

  void init(standard_context_t& context, int num_nodes, int num_edges, float queue_sizing) {
    // capacity arithmetic is the reference's: (int)(num_edges*queue_sizing) in float (enactor.hxx:22-23)
    size_t cap = (size_t)(int)(num_edges * queue_sizing);
    buffers.push_back(std::make_shared<frontier_t<int>>(context, cap));
    buffers.push_back(std::make_shared<frontier_t<int>>(context, cap));

    indices = std::make_shared<frontier_t<int>>(context, num_nodes);
    filtered_indices = std::make_shared<frontier_t<int>>(context, num_nodes);
    mem_t<int> indices_array = mgx::fill_function<int>([] __device__(int index) { return index; }, num_nodes, context);
    (void)indices->load(indices_array);
    (void)filtered_indices->load(indices_array);
    unvisited.push_back(indices);
    unvisited.push_back(filtered_indices);
  }

  enactor_t(const enactor_t& rhs) = delete;
  enactor_t& operator=(const enactor_t& rhs) = delete;
};

}  // namespace gunrock
