// ref_bench_bfs -- what the DROP-IN costs: the reference's own bfs_enactor_t::enact_pushpull, bfs_problem_t and bfs_functor_t,
// UNCHANGED (read from /root/reference through tests/dropin/build_dropin.sh's symlink farm), on this repo's advance / filter
// operators, run a few times on a graph handed over as a raw binary CSR -- warm timings, which the reference's one-shot
// test_bfs.cu cannot give.  This file is the repo's own (a driver, like tools/golden_ref/driver.cpp); it is test and
// measurement infrastructure: only its binary travels to the GPU box.
//   ref_bench_bfs <graph.bin> <src> [<src> ...]
// graph.bin: int32 n, int32 pad, int64 m, int32 row_offsets[n + 1], int32 col_indices[m]   (tools/dropin_cost.py writes it)
#include "bfs/bfs_enactor.hxx"
#include "test_utils.hxx"

#include <chrono>
#include <cstdio>
#include <cstdlib>

using namespace gunrock;
using namespace gunrock::bfs;

int main(int argc, char** argv) {
  if (argc < 3) { std::fprintf(stderr, "usage: ref_bench_bfs <graph.bin> <src> [<src> ...]\n"); return 2; }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) { std::perror(argv[1]); return 2; }
  int n = 0, pad = 0;
  long long m = 0;
  if (std::fread(&n, 4, 1, f) != 1 || std::fread(&pad, 4, 1, f) != 1 || std::fread(&m, 8, 1, f) != 1 || n <= 0 || m <= 0 || m > 2147483647LL) {
    std::fprintf(stderr, "bad header\n"); return 2;
  }
  std::shared_ptr<graph_t> graph(std::make_shared<graph_t>());
  graph->undirected = true; graph->num_nodes = n; graph->num_edges = (int)m;
  graph->csr = std::make_shared<csr_t>();
  csr_t& c = *graph->csr;
  c.num_nodes = n; c.num_edges = (int)m;
  c.offsets.resize((size_t)n + 1); c.indices.resize((size_t)m);
  if (std::fread(c.offsets.data(), 4, (size_t)n + 1, f) != (size_t)n + 1 || std::fread(c.indices.data(), 4, (size_t)m, f) != (size_t)m) {
    std::fprintf(stderr, "short file\n"); return 2;
  }
  std::fclose(f);
  c.edge_weights.assign((size_t)m, 1.0f);
  c.sources.resize((size_t)m);
  for (int v = 0; v < n; ++v)
    for (int e = c.offsets[(size_t)v]; e < c.offsets[(size_t)v + 1]; ++e) c.sources[(size_t)e] = v;
  graph->csc = graph->csr;                       // (what the reference's loader ends up with)

  standard_context_t context;
  std::shared_ptr<graph_device_t> d_graph(std::make_shared<graph_device_t>());
  graph_to_device(d_graph, graph, context);
  const float alpha = 1.0f / d_graph->num_nodes;   // test_bfs.cu:30: push only
  std::shared_ptr<bfs_enactor_t> enactor(std::make_shared<bfs_enactor_t>(context, d_graph->num_nodes, d_graph->num_edges));
  bool all_ok = true;
  for (int a = 2; a < argc; ++a) {
    const int src = std::atoi(argv[a]);
    double best = 1e30;
    long long m_t = 0;
    for (int rep = 0; rep < 3; ++rep) {
      std::shared_ptr<bfs_problem_t> problem(std::make_shared<bfs_problem_t>(d_graph, src, context));
      context.synchronize();
      const auto t0 = std::chrono::steady_clock::now();
      enactor->enact_pushpull(problem, alpha, context);
      context.synchronize();
      const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
      if (ms < best) best = ms;
      if (rep == 0) {
        std::vector<int> want((size_t)n, -1);
        problem->extract();
        problem->cpu(want, graph->csr->offsets, graph->csr->indices);
        const bool ok = validate(problem->labels, want);
        all_ok = all_ok && ok;
        m_t = 0;
        for (int v = 0; v < n; ++v) if (want[(size_t)v] >= 0) m_t += c.offsets[(size_t)v + 1] - c.offsets[(size_t)v];
        std::printf("src %d: %s\n", src, ok ? "Correct." : "Validation Error.");
      }
    }
    std::printf("RESULT src %d m_t %lld best_ms %.4f\n", src, m_t, best);
  }
  return all_ok ? 0 : 1;
}
