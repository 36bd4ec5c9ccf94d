// Golden-vector driver (tools/regen_goldens.sh): runs the REFERENCE's own loader and CPU validators -- included from
// /root/reference/gunrock/src where they lie, never copied -- on one MatrixMarket file and prints what they produce as
// one JSON object.  usage: driver <file.mtx> <undir 0|1> <src>
//   load_graph                 graph.hxx:96-223
//   bfs_problem_t::cpu         bfs/bfs_problem.hxx:52-72
//   sssp_problem_t::cpu        sssp/sssp_problem.hxx:59-88
//   kcore_problem_t::cpu       kcore/kcore_problem.hxx:54-105   (undirected loads only: test_kcore.cu:24)
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "bfs/bfs_problem.hxx"
#include "sssp/sssp_problem.hxx"
#include "kcore/kcore_problem.hxx"

using namespace gunrock;
using namespace mgpu;

template <class T>
static void dump(const char* name, const std::vector<T>& v, const char* fmt, bool last = false) {
  printf("\"%s\": [", name);
  for (size_t i = 0; i < v.size(); ++i) {
    if (i) printf(",");
    printf(fmt, v[i]);
  }
  printf("]%s\n", last ? "" : ",");
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const bool undir = atoi(argv[2]) != 0;
  const int src = atoi(argv[3]);
  standard_context_t ctx(false);
  std::shared_ptr<graph_t> g = load_graph(argv[1], undir, false);
  if (!g) return 3;
  std::shared_ptr<graph_device_t> dg(new graph_device_t());
  graph_to_device(dg, g, ctx);
  std::vector<int> labels(g->num_nodes, -1), preds(g->num_nodes, -1);
  {
    bfs::bfs_problem_t bp(dg, src, ctx);
    bp.cpu(labels, g->csr->offsets, g->csr->indices);
  }
  {
    sssp::sssp_problem_t sp(dg, src, ctx);
    sp.cpu(preds, g->csr->offsets, g->csr->indices, g->csr->edge_weights);
  }
  std::vector<int> cores(g->num_nodes, 0);   // test_kcore.cu:36
  int largest_k_core = -1;
  {
    kcore::kcore_problem_t kp(dg, ctx);
    // degrees = CSR row lengths, what GetDegrees computes (problem.hxx:23-30); see the note in moderngpu/memory.hxx
    for (int v = 0; v < g->num_nodes; ++v) kp.degrees[v] = g->csr->offsets[v + 1] - g->csr->offsets[v];
    largest_k_core = kp.cpu(cores, g->csr->offsets, g->csr->indices);
  }
  printf("{\"n\": %d, \"m\": %d, \"src\": %d, \"graph_t_undirected\": %d,\n", g->num_nodes, g->num_edges, src, (int)g->undirected);
  dump("offsets", g->csr->offsets, "%d");
  dump("indices", g->csr->indices, "%d");
  dump("weights", g->csr->edge_weights, "%.9g");
  dump("sources", g->csr->sources, "%d");
  dump("csc_offsets", g->csc->offsets, "%d");
  dump("csc_indices", g->csc->indices, "%d");
  dump("bfs_labels", labels, "%d");
  dump("kcore_num_cores", cores, "%d");
  printf("\"kcore_largest\": %d,\n", largest_k_core);
  dump("sssp_preds", preds, "%d", true);
  printf("}\n");
  return 0;
}
