// Sanitizer harness for the product's HOST-ONLY entry points: the MatrixMarket loader and the binary CSR cache
// (include/gunrock/graph.hxx: load_graph / save_graph_cache / load_graph_cache -- what mgx_load_mtx, mgx_load_mtx_csc,
// mgx_graph_save_csr and mgx_graph_load_csr wrap).  Built by tools/host_asan/run.sh with AddressSanitizer +
// UndefinedBehaviorSanitizer on the HOST side only (the GPU pool has no device ASan); no HIP call is made: it runs in the
// build container.  usage: host_io_asan <dir with the golden .mtx fixtures> <scratch dir>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "gunrock/graph.hxx"

using namespace gunrock;

static int failures = 0;
#define EXPECT(cond, what) do { if (!(cond)) { std::printf("FAIL: %s (%s:%d)\n", what, __FILE__, __LINE__); ++failures; } } while (0)

static std::vector<unsigned char> slurp(const std::string& p) {
  std::vector<unsigned char> b;
  FILE* f = fopen(p.c_str(), "rb");
  if (!f) return b;
  unsigned char buf[4096];
  size_t k;
  while ((k = fread(buf, 1, sizeof(buf), f)) > 0) b.insert(b.end(), buf, buf + k);
  fclose(f);
  return b;
}
static void spit(const std::string& p, const std::vector<unsigned char>& b, size_t len) {
  FILE* f = fopen(p.c_str(), "wb");
  if (len) fwrite(b.data(), 1, len, f);
  fclose(f);
}
static bool same(const graph_t& a, const graph_t& b) {
  auto eq = [](const csr_t& x, const csr_t& y) {
    return x.offsets == y.offsets && x.indices == y.indices && x.edge_weights == y.edge_weights && x.sources == y.sources;
  };
  return a.num_nodes == b.num_nodes && a.num_edges == b.num_edges && a.undirected == b.undirected && eq(*a.csr, *b.csr) && eq(*a.csc, *b.csc) &&
         ((a.csc == a.csr) == (b.csc == b.csr));
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const std::string gold = argv[1], tmp = argv[2];
  const char* fixtures[] = {"bfs_test.mtx", "sssp_test.mtx", "pr_test.mtx", "kcore_test.mtx", "synthetic_dup.mtx"};
  int round = 0;
  for (const char* fx : fixtures) {
    for (int undir = 0; undir < 2; ++undir)
      for (int csc = 0; csc < 2; ++csc) {
        auto g = load_graph((gold + "/" + fx).c_str(), undir != 0, false, csc != 0);
        EXPECT(g != nullptr, fx);
        if (!g) continue;
        EXPECT((int)g->csr->offsets.size() == g->num_nodes + 1 && g->csr->offsets.back() == g->num_edges, "CSR shape");
        const std::string cache = tmp + "/cache_" + std::to_string(round++) + ".bin";
        EXPECT(save_graph_cache(cache.c_str(), *g), "cache written");
        auto h = load_graph_cache(cache.c_str());
        EXPECT(h != nullptr && same(*g, *h), "cache round trip");
        // every truncation, a flipped byte at every position of a small file, a byte too many: rejected, never trusted
        const std::vector<unsigned char> bytes = slurp(cache);
        const std::string bad = tmp + "/bad.bin";
        for (size_t len = 0; len < bytes.size(); len += (bytes.size() > 600 ? 7 : 1)) {
          spit(bad, bytes, len);
          EXPECT(load_graph_cache(bad.c_str()) == nullptr, "truncated cache accepted");
        }
        for (size_t at = 0; at < bytes.size(); at += (bytes.size() > 600 ? 5 : 1)) {
          std::vector<unsigned char> b = bytes;
          b[at] ^= 0x5A;
          spit(bad, b, b.size());
          auto r = load_graph_cache(bad.c_str());
          EXPECT(r == nullptr, "corrupt cache accepted");
        }
        {
          std::vector<unsigned char> b = bytes;
          b.push_back(0);
          spit(bad, b, b.size());
          EXPECT(load_graph_cache(bad.c_str()) == nullptr, "cache with a trailing byte accepted");
        }
        // a header that promises far more than the file holds must not be believed (no allocation by its numbers)
        for (int field = 0; field < 2; ++field) {
          std::vector<unsigned char> b = bytes;
          const int huge = 0x7FFFFFF0;
          memcpy(b.data() + 8 + 4 * field, &huge, 4);
          spit(bad, b, b.size());
          EXPECT(load_graph_cache(bad.c_str()) == nullptr, "cache with an inflated header accepted");
        }
      }
  }
  // malformed MatrixMarket text
  const char* texts[] = {
      "",                                                   // empty
      "%%MatrixMarket matrix coordinate real general\n",    // no size line
      "%c\nx y z\n",                                        // size line not numeric
      "3 3 -1\n",                                           // negative count
      "3 3 2\n1 2\n",                                       // fewer entries than promised
      "3 3 1\n0 1\n",                                       // id 0 (ids are 1-based)
      "3 3 1\n1 5\n",                                       // id past the size
      "3 3 1\n-1 2\n",                                      // negative id
      "3 3 1\n1\n",                                         // one number only
      "3 3 2000000000\n1 2\n",                              // a count beyond int32 once doubled
  };
  int t = 0;
  for (const char* text : texts) {
    const std::string p = tmp + "/bad_" + std::to_string(t++) + ".mtx";
    FILE* f = fopen(p.c_str(), "w");
    fputs(text, f);
    fclose(f);
    for (int undir = 0; undir < 2; ++undir) {
      auto g = load_graph(p.c_str(), undir != 0, false, true);
      if (g) EXPECT((int)g->csr->offsets.size() == g->num_nodes + 1, "accepted text with a broken CSR");
    }
  }
  EXPECT(load_graph((tmp + "/does_not_exist.mtx").c_str()) == nullptr, "missing file");
  std::printf("host_io_asan: %d failures\n", failures);
  return failures ? 1 : 0;
}
