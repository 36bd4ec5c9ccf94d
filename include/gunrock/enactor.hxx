// gunrock/enactor.hxx -- enactor_t, the base of every superstep loop: the frontiers a loop ping-pongs between.
// Drop-in for the reference's gunrock/src/enactor.hxx:11-46 (same members, same constructor):
//   buffers[0..1]                 edge-capacity frontiers, (int)(num_edges * queue_sizing) entries each -- the float
//                                 arithmetic of enactor.hxx:22-23 is kept, frontier overflow depends on it
//   indices, filtered_indices     node-capacity frontiers pre-filled with 0 .. num_nodes-1
//   unvisited[0..1]               the same two, as the pull phase of BFS addresses them
#pragma once

#include "frontier.hxx"
#include "graph.hxx"
#include "problem.hxx"

namespace gunrock {

struct enactor_t {
  typedef std::shared_ptr<frontier_t<int>> frontier_ptr;

  std::vector<frontier_ptr> buffers;
  frontier_ptr indices;
  frontier_ptr filtered_indices;
  std::vector<frontier_ptr> unvisited;

  enactor_t(standard_context_t& context, int num_nodes, int num_edges, float queue_sizing = 1.0f) {
    init(context, num_nodes, num_edges, queue_sizing);
  }
  enactor_t(const enactor_t&) = delete;
  enactor_t& operator=(const enactor_t&) = delete;

  void init(standard_context_t& context, int num_nodes, int num_edges, float queue_sizing) {
    const size_t edge_capacity = (size_t)(int)(num_edges * queue_sizing);
    for (int i = 0; i < 2; ++i) buffers.push_back(std::make_shared<frontier_t<int>>(context, edge_capacity));

    mem_t<int> iota = mgx::fill_function<int>([] __device__(int v) { return v; }, num_nodes, context);
    for (frontier_ptr* f : {&indices, &filtered_indices}) {
      *f = std::make_shared<frontier_t<int>>(context, (size_t)num_nodes);
      (void)(*f)->load(iota);
      unvisited.push_back(*f);
    }
  }
};

}  // namespace gunrock
