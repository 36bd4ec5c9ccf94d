// mgx_capi.hip -- implementation of the C-ABI declared in include/mgx.h.
// Pre-instantiates the operator templates of include/gunrock/*.hxx for the in-scope functors
// (BFS, SSSP, PR) and exposes the building blocks.  Built with
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Iinclude
// into mini_amd/libmgx.so (see __graft_entry__.build()).
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>

#include "gunrock/bfs/bfs_enactor.hxx"
#include "gunrock/pr/pr_enactor.hxx"
#include "gunrock/kcore/kcore_enactor.hxx"
#include "gunrock/sssp/sssp_enactor.hxx"
#include "mgx/bfs_dist.hpp"
#include "mgx/bfs_dist2.hpp"
#include "mgx/env.hpp"
#include "mgx/sssp_dist.hpp"
#include "mgx/rmat.hpp"
#include "mgx/sssp_fused.hpp"
#include "mgx/sssp_preds.hpp"
#include "mgx.h"

using namespace gunrock;

// ------------------------------------------------------------------------------------------
// handles
// ------------------------------------------------------------------------------------------
struct mgx_ctx_s {
  int device;
  std::unique_ptr<standard_context_t> ctx;
};
struct mgx_graph_s {
  mgx_ctx_s* c;
  std::shared_ptr<graph_device_t> g;
  bool weights_checked = false;
  bool weights_ok = true;
};
struct mgx_frontier_s {
  mgx_ctx_s* c;
  std::shared_ptr<frontier_t<int>> f;
};
struct mgx_bfs_s {
  mgx_graph_s* g;
  std::shared_ptr<bfs::bfs_problem_t> p;
  std::unique_ptr<bfs::bfs_enactor_t> e;              // lazily: holds two m-capacity buffers
  std::unique_ptr<bfs::bfs_fused_enactor_t> fe;       // lazily: O(n)
  int64_t last_stats[24] = {0};
  mem_t<unsigned> visited_mask;                       // lazily: the idempotent mode's bitmask ((n + 31) / 32 words)
  int time_kernels = -1;                              // -1: environment default
};
struct mgx_sssp_s {
  mgx_graph_s* g;
  std::shared_ptr<sssp::sssp_problem_t> p;
  std::unique_ptr<sssp::sssp_enactor_t> e;
  float e_sizing = 0.f;
  std::unique_ptr<mgx::sssp_fused_state_t> fused;     // lazily: O(n)
  // predecessors of the fused loop's distances (mgx/sssp_preds.hpp): built when somebody asks for them
  mgx::sssp_pred_state_t pred_state;
  bool preds_stale = false;
  int preds_src = 0;
};
struct mgx_pr_s {
  mgx_graph_s* g;
  std::shared_ptr<pr::pr_problem_t> p;
  std::unique_ptr<pr::pr_enactor_t> e;
};
struct mgx_kcore_s {
  mgx_graph_t g = nullptr;
  std::shared_ptr<kcore::kcore_problem_t> p;
  std::unique_ptr<kcore::kcore_enactor_t> e;
};

struct mgx_dbfs_s {
  mgx_ctx_s* c;
  mgx::dbfs_state_t st;
};

struct mgx_dsssp_s {
  mgx_ctx_s* c;
  mgx::dsssp_state_t st;
  mgx::dsssp_run_bufs_t run_bufs;
};

struct mgx_dbfs2_s {
  mgx_ctx_s* c;
  mgx::d2_state_t st;
  mgx::d2_run_bufs_t run_bufs;
  mgx::d2_group_bufs_t group_bufs;     // (rank 0's handle holds the group run's shared buffers: mgx_dbfs2_run_group)
};
struct mgx_comm_s {
  mgx_ctx_s* c;
  mgx::comm_t cm;
};

static thread_local std::string g_last_error;

#define MGX_TRY try {
#define MGX_CATCH                                                  \
  }                                                                \
  catch (const mgx::mgx_error& e) {                                \
    g_last_error = e.what();                                       \
    return e.code;                                                 \
  }                                                                \
  catch (const mgx::hip_error& e) {                                \
    g_last_error = e.what();                                       \
    return MGX_E_HIP;                                              \
  }                                                                \
  catch (const std::exception& e) {                                \
    g_last_error = e.what();                                       \
    return MGX_E_INVALID;                                          \
  }                                                                \
  return MGX_OK;

#define MGX_REQUIRE(cond, msg)                                    \
  do {                                                             \
    if (!(cond)) throw mgx::mgx_error(MGX_E_INVALID, msg);         \
  } while (0)

// every entry point: the handle's device, and its stream as the one the context-less copies are ordered on
static inline void use_device(mgx_ctx_s* c) { MGX_HIP(hipSetDevice(c->device)); c->ctx->make_current(); }

// neighbourhood reduce with a plain per-vertex gather as the functor
namespace {
template <typename V>
struct gather_problem_t : problem_t {
  struct data_slice_t { const V* values; };
  mem_t<data_slice_t> d_data_slice;
  gather_problem_t(std::shared_ptr<graph_device_t> g, const V* values, standard_context_t& ctx) : problem_t(g) {
    std::vector<data_slice_t> h(1);
    h[0].values = values;
    d_data_slice = to_mem(h, ctx);
  }
};
template <typename V>
struct gather_functor_t {
  typedef typename gather_problem_t<V>::data_slice_t slice_t;
  static constexpr bool mgx_pure_gather = true;      // cond / apply are trivially true, the value is a pure read (neighborhood.hxx)
  static __device__ __forceinline__ bool cond_advance(int, int, int, int, int, slice_t*, int) { return true; }
  static __device__ __forceinline__ bool apply_advance(int, int, int, int, int, slice_t*, int) { return true; }
  static __device__ __forceinline__ V get_value_to_reduce(int idx, slice_t* d, int) { return d->values[idx]; }
};

extern "C" int mgx_nrs_build_device(const int* ro, const int* ci, int rows, unsigned slice_n, int slices, void** mu, unsigned** off,
                                    unsigned* first, long long* total, hipStream_t stream);   // mgx_layout.hip
// The long rows by slice of their destinations (mgx/nreduce.hpp: k_nrs_edges), once per graph at its first neighbour-reduce
// through the library: needs the layout with its degree classes (the long rows are [0, vs_v[0])); MGX_NR_SLICED=0 skips it.
// 16 bytes per mini-unit -- RMAT-22: 17.5 M of them, 280 MB -- and a partial per mini-unit in the context's arena; left out
// (the unit blocks serve) when the memory is not there.
static void ensure_nr_slices(mgx_graph_s* g) {
  graph_device_t& G = *g->g;
  if (G.nrs_tried) return;
  G.nrs_tried = true;
  if (const char* e = mgx::env("MGX_NR_SLICED")) if (atoi(e) == 0) return;
  if (!G.has_layout || G.ub_units <= 0 || !G.d_ub_first.size() || G.vs_long_min != G.ub_min_degree || G.vs_long_min < 17 || G.vs_long_min > 64 ||
      G.vs_v[0] == 0 || G.vs_dummy == 0) return;
  standard_context_t& ctx = *g->c->ctx;
  const unsigned S = (unsigned)mgx::NR_HOTV;
  const long long n = G.num_nodes;
  const int all = (int)std::min<long long>((n + S - 1) / S, (long long)mgx::NRS_MAX_SLICES);      // (slices the id range has)
  int slices = std::min(all, mgx::nrs_default_slices(n));
  if (const char* e = mgx::env("MGX_NR_SLICES")) { const int v = atoi(e); if (v >= 1) slices = std::min(v, all); }   // (tests: a tail on small graphs, many slices on mid-size ones)
  const int rows = (int)G.vs_v[0];
  void* mu = nullptr;
  unsigned* off = nullptr;
  unsigned first[mgx::NRS_MAX_SLICES + 2] = {0};
  long long total = 0;
  ctx.synchronize();
  const int rc = mgx_nrs_build_device(G.d_layout_row_offsets.data(), G.d_layout_col_indices.data(), rows, S, slices, &mu, &off, first, &total,
                                      ctx.stream());
  if (rc != 0) {                       // (the memory is not there: the unit blocks serve -- the call that asked is not failed for it)
    (void)hipGetLastError();
    if (mu) (void)hipFree(mu);
    if (off) (void)hipFree(off);
    return;
  }
  if (total <= 0 || !mu || !off) return;
  G.d_nrs_mu = mem_t<unsigned>::adopt((unsigned*)mu, ((size_t)total + 4) * 4);
  G.d_nrs_off = mem_t<unsigned>::adopt(off, (size_t)(slices + 1) * (size_t)rows + 1);
  // rows of more than NRS_BIG_DEG entries (the layout is sorted by degree: a prefix), from the first rows' offsets
  {
    std::vector<int> h((size_t)rows + 1);
    MGX_HIP(mgx::dtoh(h.data(), G.d_layout_row_offsets.data(), (size_t)rows + 1));
    auto first_at_most = [&](int d) {            // first row of at most d entries (degrees are non-increasing)
      size_t lo = 0, hi = (size_t)rows;
      while (lo < hi) { const size_t mid = (lo + hi) / 2; if (h[mid + 1] - h[mid] > d) lo = mid + 1; else hi = mid; }
      return (unsigned)lo;
    };
    int degs[3] = {mgx::NRS_FOLD_DEG[0], mgx::NRS_FOLD_DEG[1], mgx::NRS_FOLD_DEG[2]};
    if (const char* e = mgx::env("MGX_NR_FOLD_DEGS")) {            // (tests: every tier on small graphs)
      int d0 = 0, d1 = 0, d2 = 0;
      if (sscanf(e, "%d/%d/%d", &d0, &d1, &d2) == 3 && d0 >= d1 && d1 >= d2 && d2 >= 0) { degs[0] = d0; degs[1] = d1; degs[2] = d2; }
    }
    unsigned prev = 0;
    for (int i = 0; i < 3; ++i) { G.nrs_tier[i] = std::max(prev, first_at_most(degs[i])); prev = G.nrs_tier[i]; }
  }
  try {
    ctx.reserve_scratch(mgx::nr_scratch_bytes(G.num_nodes, total, 8));    // (a partial per mini-unit)
  } catch (const mgx::mgx_error&) {                                       // no room for the partials: give the slices back, the unit blocks serve
    (void)hipGetLastError();
    G.d_nrs_mu = mem_t<unsigned>();
    G.d_nrs_off = mem_t<unsigned>();
    return;
  }
  for (int k = 0; k < mgx::NRS_MAX_SLICES + 2; ++k) G.nrs_first[k] = k <= slices + 1 ? first[k] : first[slices + 1];
  G.nrs_slices = (unsigned)slices; G.nrs_rows = (unsigned)rows; G.nrs_units = total;
}

template <typename V, typename Op>
int segreduce_impl(mgx_graph_t g, mgx_frontier_t in, int push, const V* vals, V identity, V* reduced, int64_t* nz) {
  MGX_TRY
  MGX_REQUIRE(g && in && vals && reduced, "segreduce: NULL argument");
  use_device(g->c);
  standard_context_t& ctx = *g->c->ctx;
  if (in->f && (long long)in->f->size() * 8 >= (long long)g->g->num_nodes) ensure_nr_slices(g);     // (a full frontier, or a large ascending subset, may take mgx/nreduce.hpp)
  auto prob = std::make_shared<gather_problem_t<V>>(g->g, vals, ctx);
  std::shared_ptr<frontier_t<int>> dummy;
  int r;
  if (push)
    r = oprtr::neighborhood::neighborhood_kernel<gather_problem_t<V>, gather_functor_t<V>, V, Op, false, true>(
        prob, in->f, dummy, reduced, identity, 0, ctx);
  else
    r = oprtr::neighborhood::neighborhood_kernel<gather_problem_t<V>, gather_functor_t<V>, V, Op, false, false>(
        prob, in->f, dummy, reduced, identity, 0, ctx);
  ctx.synchronize();
  if (nz) *nz = r;
  MGX_CATCH
}
}  // namespace


extern "C" {

int mgx_version(void) { return 100; }

const char* mgx_strerror(int status) {
  switch (status) {
    case MGX_OK: return "ok";
    case MGX_E_INVALID: return "invalid argument";
    case MGX_E_HIP: return "HIP runtime error";
    case MGX_E_FRONTIER_OVERFLOW: return "frontier capacity overflow";
    case MGX_E_NEGATIVE_WEIGHT: return "negative edge weight";
    case MGX_E_NO_DEVICE: return "no HIP device";
    default: return "unknown status";
  }
}
const char* mgx_last_error(void) { return g_last_error.c_str(); }

// ---- context -------------------------------------------------------------------------------
int mgx_ctx_create(int device, void* stream, mgx_ctx_t* out) {
  MGX_TRY
  MGX_REQUIRE(out, "mgx_ctx_create: out is NULL");
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    g_last_error = "no HIP device visible";
    return MGX_E_NO_DEVICE;
  }
  MGX_REQUIRE(device >= 0 && device < count, "mgx_ctx_create: device ordinal out of range");
  MGX_HIP(hipSetDevice(device));
  auto* c = new mgx_ctx_s();
  c->device = device;
  c->ctx.reset(new standard_context_t(false, (hipStream_t)stream));
  *out = c;
  MGX_CATCH
}
int mgx_ctx_set_stream(mgx_ctx_t c, void* stream) {
  MGX_TRY
  MGX_REQUIRE(c, "ctx is NULL");
  c->ctx->set_stream((hipStream_t)stream);
  MGX_CATCH
}
int mgx_ctx_synchronize(mgx_ctx_t c) {
  MGX_TRY
  MGX_REQUIRE(c, "ctx is NULL");
  use_device(c);
  c->ctx->synchronize();
  MGX_CATCH
}
int mgx_ctx_destroy(mgx_ctx_t c) {
  MGX_TRY
  if (c) { use_device(c); delete c; }
  MGX_CATCH
}
int mgx_ctx_num_cus(mgx_ctx_t c, int* out) {
  MGX_TRY
  MGX_REQUIRE(c && out, "NULL argument");
  *out = c->ctx->num_cus;
  MGX_CATCH
}

// ---- graph ---------------------------------------------------------------------------------
static void finish_graph(mgx_ctx_s* c, graph_device_t& g) {
  g.d_scanned_row_offsets = mem_t<int>((size_t)g.num_nodes + 1, *c->ctx);
  c->ctx->reserve_scratch((size_t)g.num_edges / 2 + ((size_t)g.num_nodes + 4096) * 2 + (1 << 20));
}

int mgx_graph_upload(mgx_ctx_t c, int n, int64_t m, const int* ro, const int* ci, const float* w, const int* co,
                     const int* ri, const float* rw, mgx_graph_t* out) {
  MGX_TRY
  MGX_REQUIRE(c && out && ro && (ci || m == 0), "mgx_graph_upload: NULL argument");
  MGX_REQUIRE(n >= 0 && m >= 0 && m <= 2147483647LL, "mgx_graph_upload: sizes must fit int32 (graph.hxx:19-26)");
  MGX_REQUIRE((co == nullptr) == (ri == nullptr), "mgx_graph_upload: col_offsets and row_indices go together");
  // host arrays: cheap to validate (the kernels index with them unchecked)
  MGX_REQUIRE(ro[0] == 0 && (int64_t)ro[n] == m, "mgx_graph_upload: row_offsets must start at 0 and end at num_edges");
  for (int v = 0; v < n; ++v) MGX_REQUIRE(ro[v] <= ro[v + 1], "mgx_graph_upload: row_offsets must be non-decreasing");
  for (int64_t e = 0; e < m; ++e) MGX_REQUIRE(ci[e] >= 0 && ci[e] < n, "mgx_graph_upload: col_indices outside [0, num_nodes)");
  if (co) {
    MGX_REQUIRE(co[0] == 0 && (int64_t)co[n] == m, "mgx_graph_upload: col_offsets must start at 0 and end at num_edges");
    for (int v = 0; v < n; ++v) MGX_REQUIRE(co[v] <= co[v + 1], "mgx_graph_upload: col_offsets must be non-decreasing");
    for (int64_t e = 0; e < m; ++e) MGX_REQUIRE(ri[e] >= 0 && ri[e] < n, "mgx_graph_upload: row_indices outside [0, num_nodes)");
  }
  use_device(c);
  auto g = std::make_shared<graph_device_t>();
  g->num_nodes = n;
  g->num_edges = (int)m;
  g->d_row_offsets = mem_t<int>((size_t)n + 1, *c->ctx);
  MGX_HIP(mgx::htod(g->d_row_offsets.data(), ro, (size_t)n + 1));
  g->d_col_indices = mem_t<int>((size_t)m, *c->ctx);
  MGX_HIP(mgx::htod(g->d_col_indices.data(), ci, (size_t)m));
  if (w) {
    g->d_col_values = mem_t<float>((size_t)m, *c->ctx);
    MGX_HIP(mgx::htod(g->d_col_values.data(), w, (size_t)m));
  } else {
    g->d_col_values = mgx::fill(1.0f, (size_t)m, *c->ctx);
  }
  if (co) {
    g->d_col_offsets = mem_t<int>((size_t)n + 1, *c->ctx);
    MGX_HIP(mgx::htod(g->d_col_offsets.data(), co, (size_t)n + 1));
    g->d_row_indices = mem_t<int>((size_t)m, *c->ctx);
    MGX_HIP(mgx::htod(g->d_row_indices.data(), ri, (size_t)m));
    if (rw) {
      g->d_row_values = mem_t<float>((size_t)m, *c->ctx);
      MGX_HIP(mgx::htod(g->d_row_values.data(), rw, (size_t)m));
    } else {
      g->d_row_values = mgx::fill(1.0f, (size_t)m, *c->ctx);
    }
    g->csc_is_csr = false;
  } else {
    g->d_col_offsets = mem_t<int>::borrow(g->d_row_offsets.data(), g->d_row_offsets.size());
    g->d_row_indices = mem_t<int>::borrow(g->d_col_indices.data(), g->d_col_indices.size());
    g->d_row_values = mem_t<float>::borrow(g->d_col_values.data(), g->d_col_values.size());
    g->csc_is_csr = true;
  }
  finish_graph(c, *g);
  c->ctx->synchronize();
  auto* h = new mgx_graph_s();
  h->c = c;
  h->g = g;
  *out = h;
  MGX_CATCH
}

int mgx_graph_wrap_device(mgx_ctx_t c, int n, int64_t m, const int* ro, const int* ci, const float* w, const int* co,
                          const int* ri, const float* rw, mgx_graph_t* out) {
  MGX_TRY
  MGX_REQUIRE(c && out && ro && (ci || m == 0), "mgx_graph_wrap_device: NULL argument");
  MGX_REQUIRE(n >= 0 && m >= 0 && m <= 2147483647LL, "mgx_graph_wrap_device: sizes must fit int32");
  MGX_REQUIRE((co == nullptr) == (ri == nullptr), "mgx_graph_wrap_device: col_offsets and row_indices go together");
  use_device(c);
  auto g = std::make_shared<graph_device_t>();
  g->num_nodes = n;
  g->num_edges = (int)m;
  g->d_row_offsets = mem_t<int>::borrow((int*)ro, (size_t)n + 1);
  g->d_col_indices = mem_t<int>::borrow((int*)ci, (size_t)m);
  if (w) g->d_col_values = mem_t<float>::borrow((float*)w, (size_t)m);
  else g->d_col_values = mgx::fill(1.0f, (size_t)m, *c->ctx);
  if (co) {
    g->d_col_offsets = mem_t<int>::borrow((int*)co, (size_t)n + 1);
    g->d_row_indices = mem_t<int>::borrow((int*)ri, (size_t)m);
    if (rw) g->d_row_values = mem_t<float>::borrow((float*)rw, (size_t)m);
    else g->d_row_values = mgx::fill(1.0f, (size_t)m, *c->ctx);
    g->csc_is_csr = false;
  } else {
    g->d_col_offsets = mem_t<int>::borrow(g->d_row_offsets.data(), (size_t)n + 1);
    g->d_row_indices = mem_t<int>::borrow(g->d_col_indices.data(), (size_t)m);
    g->d_row_values = mem_t<float>::borrow(g->d_col_values.data(), (size_t)m);
    g->csc_is_csr = true;
  }
  finish_graph(c, *g);
  c->ctx->synchronize();
  auto* h = new mgx_graph_s();
  h->c = c;
  h->g = g;
  *out = h;
  MGX_CATCH
}
static void build_unit_blocks(mgx_graph_s* g);
int mgx_graph_attach_layout(mgx_graph_t g, const int* d_row_offsets, const int* d_col_indices, const int* d_new_of_old,
                            const int* d_old_of_new) {
  MGX_TRY
  MGX_REQUIRE(g && d_row_offsets && (d_col_indices || g->g->num_edges == 0) && d_new_of_old && d_old_of_new,
              "mgx_graph_attach_layout: NULL argument");
  graph_device_t& G = *g->g;
  G.d_layout_row_offsets = mem_t<int>::borrow((int*)d_row_offsets, (size_t)G.num_nodes + 1);
  G.d_layout_col_indices = mem_t<int>::borrow((int*)d_col_indices, (size_t)G.num_edges);
  G.d_new_of_old = mem_t<int>::borrow((int*)d_new_of_old, (size_t)G.num_nodes);
  G.d_old_of_new = mem_t<int>::borrow((int*)d_old_of_new, (size_t)G.num_nodes);
  G.has_layout = true;
  G.vs_edges = 0; G.vs_dummy = 0; G.vs_long_min = 0;      // (borrowed arrays: no padding behind them, sortedness not checked)
  G.d_cold_owner = mem_t<int>(); G.d_cold_dst = mem_t<int>(); G.d_colds_owner = mem_t<int>(); G.d_colds_dst = mem_t<int>();
  G.d_cold_pk = mem_t<unsigned>(); G.d_cold_cbase = mem_t<unsigned>(); G.cold_pk_mask = 0;
  G.cold_pairs = G.colds_pairs = 0; G.cold_slices = 0;
  use_device(g->c);
  g->c->ctx->synchronize();
  build_unit_blocks(g);
  MGX_CATCH
}
int mgx_graph_attach_layout_weights(mgx_graph_t g, const float* d_layout_weights) {
  MGX_TRY
  MGX_REQUIRE(g && d_layout_weights, "mgx_graph_attach_layout_weights: NULL argument");
  MGX_REQUIRE(g->g->has_layout, "mgx_graph_attach_layout_weights: attach the layout first");
  graph_device_t& G = *g->g;
  G.d_layout_col_values = mem_t<float>::borrow((float*)d_layout_weights, (size_t)G.num_edges);
  G.has_layout_weights = true;
  MGX_CATCH
}
extern "C" int mgx_layout_build_device(const int* ro, const int* ci, const float* w, int n, long long m, int* lro, int* lci,
                                       float* lw, int* new_of_old, int* old_of_new, hipStream_t stream);   // mgx_layout.hip
extern "C" int mgx_units_build_device(const int* ro, const int* ci, int n, int min_deg, int max_deg, int ushift, unsigned hot_limit,
                                      int** owner, int** ucol, unsigned char** ucnt, int** ufirst, long long* units, long long* units_pad,
                                      hipStream_t stream);

// Unit blocks of the layout's long rows (mgx/bfs_fused_dense.hpp).  The threshold is the
// fused traversal's long-row threshold at build time (MGX_BFS_LONG_MIN, default mgx::LONG_MIN_DEFAULT; a unit is 64 entries whatever the
// threshold); a run with another threshold ignores the blocks.
namespace {
// four 32-bit entries -> three words of 24-bit entries (little endian: entry k occupies bits [24 k, 24 k + 24) of the 96)
__global__ __launch_bounds__(256) void k_pack24(const int4* __restrict__ in, long long quads, unsigned* __restrict__ out) {
  for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < quads; q += (long long)gridDim.x * blockDim.x) {
    const int4 e = in[q];
    const unsigned a = (unsigned)e.x & 0xFFFFFFu, b = (unsigned)e.y & 0xFFFFFFu, c = (unsigned)e.z & 0xFFFFFFu, d = (unsigned)e.w & 0xFFFFFFu;
    out[3 * q + 0] = a | (b << 24);
    out[3 * q + 1] = (b >> 8) | (c << 16);
    out[3 * q + 2] = (c >> 16) | (d << 8);
  }
}
}  // namespace
static void build_unit_blocks(mgx_graph_s* g) {
  graph_device_t& G = *g->g;
  G.d_ub_col = mem_t<int>(); G.d_ub_owner = mem_t<int>(); G.ub_units = G.ub_units_pad = 0; G.ub_min_degree = 0;
  G.d_ub_col24 = mem_t<unsigned>();
  G.d_ub_cnt = mem_t<unsigned char>(); G.d_ub_first = mem_t<int>(); G.nr_big_rows = 0; G.d_ub_w = mem_t<float>(); G.d_ub_w16 = mem_t<unsigned short>(); G.ub_w_tried = false;
  G.d_nrs_mu = mem_t<unsigned>(); G.d_nrs_off = mem_t<unsigned>(); G.nrs_units = 0; G.nrs_slices = G.nrs_rows = 0; G.nrs_tier[0] = G.nrs_tier[1] = G.nrs_tier[2] = 0; G.nrs_tried = false;
  int long_min = mgx::LONG_MIN_DEFAULT;
  if (const char* e = mgx::env("MGX_BFS_LONG_MIN")) long_min = atoi(e);
  if (long_min <= 0 || !G.has_layout || G.num_edges <= 0) return;
  int *owner = nullptr, *ucol = nullptr, *ufirst = nullptr;
  unsigned char* ucnt = nullptr;
  long long units = 0, units_pad = 0;
  const int rc = mgx_units_build_device(G.d_layout_row_offsets.data(), G.d_layout_col_indices.data(), G.num_nodes, long_min,
                                        0x7FFFFFFF, 6, 0u, &owner, &ucol, &ucnt, &ufirst, &units, &units_pad, g->c->ctx->stream());
  if (rc != 0) throw mgx::mgx_error(MGX_E_HIP, std::string("unit blocks: ") + hipGetErrorString((hipError_t)rc));
  if (units <= 0) return;
  G.d_ub_owner = mem_t<int>::adopt(owner, (size_t)units_pad);
  G.d_ub_col = mem_t<int>::adopt(ucol, ((size_t)units_pad << 6) + 4);
  G.d_ub_cnt = mem_t<unsigned char>::adopt(ucnt, (size_t)units_pad + 16);
  G.d_ub_first = mem_t<int>::adopt(ufirst, (size_t)G.num_nodes + 1);
  // the neighbour-reduce over the unit blocks (mgx/nreduce.hpp) keeps its values and per-unit partials in the context's arena
  g->c->ctx->reserve_scratch(mgx::nr_scratch_bytes(G.num_nodes, units_pad, 8));
  G.ub_units = units; G.ub_units_pad = units_pad; G.ub_min_degree = long_min;
  // 24-bit copy for the fused BFS (ids below 2^23: bit 23 of an entry is free, so a sign-extending unpack turns 0xFFFFFF
  // back into -1); (units_pad * 64 + 4) entries -- the four -1 behind the blocks included -- are whole quads
  bool pack = (long long)G.num_nodes <= (1ll << 23);
  if (const char* e = mgx::env("MGX_BFS_PACK24")) pack = pack && atoi(e) != 0;
  if (pack) {
    const long long quads = ((long long)units_pad << 4) + 1;
    G.d_ub_col24 = mem_t<unsigned>((size_t)quads * 3 + 4, *g->c->ctx);
    hipLaunchKernelGGL(k_pack24, dim3(mgx::grid_for(quads, 256, 16384)), dim3(256), 0, g->c->ctx->stream(), (const int4*)G.d_ub_col.data(), quads,
                       G.d_ub_col24.data());
    // Round 6 (memory): every reader of the unit blocks takes the 24-bit copy when there is one -- the fused BFS, the neighbour-reduce,
    // the fused SSSP's sweep (with half or float weights) -- so the 32-bit entries go: 493 MB of RMAT-22's 2.03 GB of layout.
    // (MGX_BFS_PACK24=0 at layout time keeps them and builds no copy.)
    g->c->ctx->synchronize();
    G.d_ub_col = mem_t<int>();
  }
}
extern "C" int mgx_cold_build_device(const int* ro, const int* ci, int n, int row0, int rows, int min_deg, unsigned hot_n, unsigned slice_n,
                                     int slices, int** owner, int** dst, long long* pairs, int* slice_off, hipStream_t stream);
extern "C" int mgx_cold_pack_device(const int* owner, const int* dst, int used, const unsigned* off, const unsigned* lo, int ranks,
                                    unsigned** pk, unsigned** cbase, unsigned* cb_off, unsigned long long* mask, hipStream_t stream);
// Cold-edge lists of the layout (mgx/bfs_fused_cold.hpp); MGX_BFS_COLD_LISTS=0 skips them.  Needs the unit blocks and the
// degree classes (a degree-sorted layout: the long rows are [0, vs_v[0]), the short ones [vs_v[0], vs_v[3])); built only
// when the cold entries are a small share of the long rows' entries (a skewed graph under the hub-first order) and few
// slices hold any.  One list for the long rows, one for the short rows, the same slices.
static void build_cold_lists(mgx_graph_s* g) {
  graph_device_t& G = *g->g;
  G.d_cold_owner = mem_t<int>(); G.d_cold_dst = mem_t<int>(); G.d_colds_owner = mem_t<int>(); G.d_colds_dst = mem_t<int>();
  G.cold_pairs = G.colds_pairs = 0; G.cold_slices = 0; G.cold_hot_n = 0; G.cold_long_min = 0; G.cold_majority = false; G.cold_all = false;
  G.d_ubh_col24 = mem_t<unsigned>(); G.d_ubh_owner = mem_t<int>(); G.ubh_units = G.ubh_units_pad = 0;
  // the short rows' list too on graphs of more than 2^23 vertices (equal to marking those entries on RMAT-22, where 6 % of the entries
  // are cold; RMAT-24 -2 %, RMAT-25 -9 % of a traversal); MGX_BFS_COLD_LISTS: 0 no lists at all, 1 the long rows' only, 2 both
  bool with_short = (long long)G.num_nodes > (1ll << 23);
  if (const char* e = mgx::env("MGX_BFS_COLD_LISTS")) {
    if (atoi(e) == 0) return;
    with_short = atoi(e) == 2;
  }
  if (G.ub_units <= 0 || G.vs_long_min <= 0 || G.vs_long_min != G.ub_min_degree || G.vs_v[0] == 0) return;
  const unsigned hot_n = (unsigned)mgx::BFS_COLD_WORDS * 32u, slice_n = hot_n;
  const unsigned n = (unsigned)G.num_nodes;
  if (n <= hot_n) return;                                          // everything is inside the prefix
  const long long slices_ll = ((long long)n - hot_n + slice_n - 1) / slice_n;
  if (slices_ll > 64) return;
  int slices = (int)slices_ll;
  unsigned list_hot_n = hot_n;          // first vertex the lists cover (0: a FLAT graph's lists hold every entry, below)
  bool flat = false;
  std::vector<int> off_l((size_t)slices + 1, 0), off_s((size_t)slices + 1, 0);
  int *owner = nullptr, *dst = nullptr;
  long long pairs = 0;
  int rc = mgx_cold_build_device(G.d_layout_row_offsets.data(), G.d_layout_col_indices.data(), (int)n, 0, (int)G.vs_v[0],
                                 G.vs_long_min, hot_n, slice_n, slices, &owner, &dst, &pairs, off_l.data(), g->c->ctx->stream());
  if (rc != 0) throw mgx::mgx_error(MGX_E_HIP, std::string("cold-edge lists: ") + hipGetErrorString((hipError_t)rc));
  if (pairs <= 0) return;
  mem_t<int> d_owner = mem_t<int>::adopt(owner, (size_t)pairs + 256), d_dst = mem_t<int>::adopt(dst, (size_t)pairs + 256);
  const long long long_entries = (long long)G.ub_units * 64;       // (padded: an upper bound of the long rows' entries)
  if (pairs * 4 > long_entries) {
    // A FLAT graph (a uniform random graph: six entries in seven point behind the LDS prefix).  Round 5 left it to the bodies that PROBE
    // the bitmap in L2 -- 134 M probes at what the L2s deliver, 1.34 ms per RMAT-22-sized traversal, whatever the kernel around them
    // does.  Round 6: EVERY entry of every row as a pair by slice of its destination, slices from vertex 0 on (the first is the LDS
    // prefix itself); a level that holds an eighth of the graph's entries is then ONE sweep of the packed pairs by the cold-edge
    // pass's workgroups, each with its slice of the bitmap in LDS -- no probe leaves the compute unit, no mark is stored -- and the
    // other levels walk their queues and mark untested (few entries: few marks).  MGX_BFS_FLAT_LISTS=0: the probes, as in round 5.
    G.cold_majority = true;
    bool want = (long long)G.num_edges < (1ll << 31) - 512 && (n + (long long)slice_n - 1) / slice_n <= 64;
    if (const char* e = mgx::env("MGX_BFS_FLAT_LISTS")) want = want && atoi(e) != 0;
    if (!want || G.vs_v[3] == 0) return;
    d_owner = mem_t<int>(); d_dst = mem_t<int>();                    // (the long rows' cold entries: superseded)
    slices = (int)((n + (long long)slice_n - 1) / slice_n);
    off_l.assign((size_t)slices + 1, 0); off_s.assign((size_t)slices + 1, 0);
    owner = nullptr; dst = nullptr; pairs = 0;
    rc = mgx_cold_build_device(G.d_layout_row_offsets.data(), G.d_layout_col_indices.data(), (int)n, 0, (int)G.vs_v[3], 1, 0u, slice_n, slices,
                               &owner, &dst, &pairs, off_l.data(), g->c->ctx->stream());
    if (rc != 0) { (void)hipGetLastError(); return; }                // (no memory for 8 bytes per entry: the probes serve)
    if (pairs <= 0) return;
    d_owner = mem_t<int>::adopt(owner, (size_t)pairs + 256); d_dst = mem_t<int>::adopt(dst, (size_t)pairs + 256);
    list_hot_n = 0u; flat = true; with_short = false;
    G.cold_majority = false;
  }
  // the short rows' cold entries
  int *owner_s = nullptr, *dst_s = nullptr;
  long long pairs_s = 0;
  mem_t<int> d_owner_s, d_dst_s;
  if (with_short && G.vs_v[3] > G.vs_v[0]) {
    rc = mgx_cold_build_device(G.d_layout_row_offsets.data(), G.d_layout_col_indices.data(), (int)n, (int)G.vs_v[0],
                               (int)(G.vs_v[3] - G.vs_v[0]), 1, hot_n, slice_n, slices, &owner_s, &dst_s, &pairs_s, off_s.data(),
                               g->c->ctx->stream());
    if (rc != 0) throw mgx::mgx_error(MGX_E_HIP, std::string("cold-edge lists: ") + hipGetErrorString((hipError_t)rc));
    if (pairs_s > 0) {
      d_owner_s = mem_t<int>::adopt(owner_s, (size_t)pairs_s + 256); d_dst_s = mem_t<int>::adopt(dst_s, (size_t)pairs_s + 256);
      if (pairs_s * 2 > (long long)G.vs_edges) { d_owner_s = mem_t<int>(); d_dst_s = mem_t<int>(); pairs_s = 0; }    // (mostly cold: leave them to the marks)
    }
  }
  if (pairs_s <= 0) std::fill(off_s.begin(), off_s.end(), 0);
  // the slices that hold pairs of either list; give up when there are too many of them
  int used = 0;
  for (int k = 0; k < slices; ++k) if (off_l[k + 1] > off_l[k] || off_s[k + 1] > off_s[k]) ++used;
  if (used > mgx::BFS_COLD_MAX_SLICES) return;
  int q = 0;
  for (int k = 0; k < slices; ++k) {
    if (!(off_l[k + 1] > off_l[k] || off_s[k + 1] > off_s[k])) continue;
    G.cold_lo[q] = list_hot_n + (unsigned)k * slice_n;
    G.cold_off[q] = (unsigned)off_l[k]; G.cold_off[q + 1] = (unsigned)off_l[k + 1];
    G.colds_off[q] = (unsigned)off_s[k]; G.colds_off[q + 1] = (unsigned)off_s[k + 1];
    ++q;
  }
  // workgroups per slice: in proportion to its pairs, at least one each; in all one per 131 072 pairs, 64 .. 1024
  const long long all = pairs + pairs_s;
  long long nwg = (all + 131071) / 131072;
  nwg = std::max<long long>(nwg, mgx::BFS_COLD_WGS);
  nwg = std::min<long long>(nwg, mgx::BFS_COLD_WGS_MAX);
  nwg = std::max<long long>(nwg, used);
  unsigned left = (unsigned)nwg - (unsigned)used, acc = 0;
  G.cold_wgs[0] = 0;
  for (int i = 0; i < used; ++i) {
    const long long cnt = ((long long)G.cold_off[i + 1] - (long long)G.cold_off[i]) + ((long long)G.colds_off[i + 1] - (long long)G.colds_off[i]);
    unsigned extra = (unsigned)((cnt * (long long)((unsigned)nwg - (unsigned)used)) / all);
    if (extra > left) extra = left;
    left -= extra;
    acc += 1u + extra;
    G.cold_wgs[i + 1] = acc;
  }
  for (int i = used + 1; i <= mgx::BFS_COLD_MAX_SLICES; ++i) { G.cold_wgs[i] = acc; G.cold_off[i] = G.cold_off[used]; G.colds_off[i] = G.colds_off[used]; }
  G.d_cold_owner = std::move(d_owner); G.d_cold_dst = std::move(d_dst);
  G.d_colds_owner = std::move(d_owner_s); G.d_colds_dst = std::move(d_dst_s);
  {
    // the long rows' pairs once more, four bytes each
    unsigned *pk = nullptr, *cbase = nullptr;
    unsigned long long mask = 0;
    const int rcp = mgx_cold_pack_device(G.d_cold_owner.data(), G.d_cold_dst.data(), used, G.cold_off, G.cold_lo, 1, &pk, &cbase, G.cold_cb, &mask,
                                         g->c->ctx->stream());
    if (rcp != 0) throw mgx::mgx_error(MGX_E_HIP, std::string("cold-edge lists, packed copy: ") + hipGetErrorString((hipError_t)rcp));
    if (pk && cbase) {
      G.d_cold_pk = mem_t<unsigned>::adopt(pk, (size_t)pairs + 256);
      G.d_cold_cbase = mem_t<unsigned>::adopt(cbase, (size_t)G.cold_cb[used] + 64);
      G.cold_pk_mask = mask;
      // Round 6 (memory): with EVERY slice packed nobody reads the 8-byte pairs of the long rows again (bfs_cold_body takes the
      // packed words slice by slice): they go -- 53 MB on RMAT-22.  (MGX_BFS_COLD_PACK=0 at layout time keeps them and packs nothing.)
      const unsigned long long every = used >= 64 ? ~0ull : ((1ull << used) - 1ull);
      if ((mask & every) == every) {
        g->c->ctx->synchronize();
        G.d_cold_owner = mem_t<int>(); G.d_cold_dst = mem_t<int>();
      }
    }
  }
  G.cold_pairs = pairs; G.colds_pairs = pairs_s; G.cold_slices = used; G.cold_hot_n = list_hot_n; G.cold_long_min = G.vs_long_min;
  G.cold_all = flat;
  if (flat && (!G.d_cold_pk.size() || G.d_cold_owner.size())) {        // (a flat graph's lists are only worth their bytes when EVERY slice is packed: 4 per entry, what the CSR costs)
    G.d_cold_owner = mem_t<int>(); G.d_cold_dst = mem_t<int>(); G.d_cold_pk = mem_t<unsigned>(); G.d_cold_cbase = mem_t<unsigned>();
    G.cold_pairs = 0; G.cold_slices = 0; G.cold_all = false; G.cold_majority = true;
    return;
  }
  if (flat) return;                          // (no blocks "without the lists' entries": the lists hold everything)
  // The unit blocks once more for the fused BFS, WITHOUT the entries that now live in the lists (its unit-block body reads them only to
  // skip them) -- and what is left points into the LDS prefix, ids below 2^20: 24 bits per entry do at every graph size (the full
  // blocks' 24-bit copy stops at 2^23 vertices).  MGX_BFS_HOT_UNITS=0: not built.  The full blocks stay: the neighbour-reduce and the
  // fused SSSP read them, and so does a traversal that is told to run without the cold-edge pass.
  bool hot_units = true;
  if (const char* e = mgx::env("MGX_BFS_HOT_UNITS")) hot_units = atoi(e) != 0;
  if (const char* e = mgx::env("MGX_BFS_PACK24")) hot_units = hot_units && atoi(e) != 0;
  if (hot_units) {
    int *owner2 = nullptr, *ucol2 = nullptr, *ufirst2 = nullptr;
    unsigned char* ucnt2 = nullptr;
    long long U2 = 0, Up2 = 0;
    const int rc3 = mgx_units_build_device(G.d_layout_row_offsets.data(), G.d_layout_col_indices.data(), G.num_nodes, G.vs_long_min, 0x7FFFFFFF, 6,
                                           hot_n, &owner2, &ucol2, &ucnt2, &ufirst2, &U2, &Up2, g->c->ctx->stream());
    if (ucnt2) (void)hipFree(ucnt2);
    if (ufirst2) (void)hipFree(ufirst2);
    if (rc3 != 0) {
      if (owner2) (void)hipFree(owner2);
      if (ucol2) (void)hipFree(ucol2);
      throw mgx::mgx_error(MGX_E_HIP, std::string("unit blocks of the hot entries: ") + hipGetErrorString((hipError_t)rc3));
    }
    if (U2 > 0) {
      mem_t<int> d_owner2 = mem_t<int>::adopt(owner2, (size_t)Up2);
      mem_t<int> d_col2 = mem_t<int>::adopt(ucol2, ((size_t)Up2 << 6) + 4);          // (freed below: only the 24-bit copy is kept)
      const long long quads = ((long long)Up2 << 4) + 1;
      G.d_ubh_col24 = mem_t<unsigned>((size_t)quads * 3 + 4, *g->c->ctx);
      hipLaunchKernelGGL(k_pack24, dim3(mgx::grid_for(quads, 256, 16384)), dim3(256), 0, g->c->ctx->stream(), (const int4*)d_col2.data(), quads,
                         G.d_ubh_col24.data());
      g->c->ctx->synchronize();
      G.d_ubh_owner = std::move(d_owner2);
      G.ubh_units = U2; G.ubh_units_pad = Up2;
    } else {
      if (owner2) (void)hipFree(owner2);
      if (ucol2) (void)hipFree(ucol2);
    }
  }
}
int mgx_graph_build_layout(mgx_graph_t g, int with_weights) {
  MGX_TRY
  MGX_REQUIRE(g, "graph is NULL");
  use_device(g->c);
  standard_context_t& ctx = *g->c->ctx;
  graph_device_t& G = *g->g;
  const size_t n = (size_t)G.num_nodes, m = (size_t)G.num_edges;
  MGX_REQUIRE(!with_weights || G.d_col_values.size() >= m, "mgx_graph_build_layout: the graph has no weights");
  ctx.synchronize();
  mem_t<int> lro(n + 1, ctx), lci(m + 8, ctx), n2o(n, ctx), o2n(n, ctx);     // (+8: slack and four -1 for bfs_fused_vshort.hpp)
  MGX_HIP(hipMemsetAsync(lci.data() + m, 0xFF, 8 * sizeof(int), ctx.stream()));
  mem_t<float> lw;
  if (with_weights) lw = mem_t<float>(m + 8, ctx);          // (+8: the short rows' 16-byte weight loads of sssp_dense_short may reach past the last entry)
  const int rc = mgx_layout_build_device(G.d_row_offsets.data(), G.d_col_indices.data(),
                                         with_weights ? G.d_col_values.data() : (const float*)nullptr, (int)n, (long long)m,
                                         lro.data(), lci.data(), with_weights ? lw.data() : (float*)nullptr, n2o.data(),
                                         o2n.data(), ctx.stream());
  if (rc != 0) throw mgx::mgx_error(MGX_E_HIP, std::string("mgx_graph_build_layout: ") + hipGetErrorString((hipError_t)rc));
  G.d_layout_row_offsets = std::move(lro);
  G.d_layout_col_indices = std::move(lci);
  G.d_new_of_old = std::move(n2o);
  G.d_old_of_new = std::move(o2n);
  G.has_layout = true;
  if (with_weights) { G.d_layout_col_values = std::move(lw); G.has_layout_weights = true; }
  build_unit_blocks(g);
  // degree classes of the short rows (the layout is sorted by degree): boundaries by binary search on a host copy
  G.vs_edges = 0; G.vs_dummy = 0; G.vs_long_min = 0;
  {
    int long_min = mgx::LONG_MIN_DEFAULT;
    if (const char* e = mgx::env("MGX_BFS_LONG_MIN")) long_min = atoi(e);
    if (long_min > 0 && long_min <= 64 && n > 0 && m > 0) {
      std::vector<int> h(n + 1);
      MGX_HIP(mgx::dtoh(h.data(), G.d_layout_row_offsets.data(), n + 1));
      auto first_below = [&](int d) {            // first vertex with degree < d (degrees are non-increasing)
        size_t lo = 0, hi = n;
        while (lo < hi) { const size_t mid = (lo + hi) / 2; if (h[mid + 1] - h[mid] >= d) lo = mid + 1; else hi = mid; }
        return (unsigned)lo;
      };
      const unsigned b0 = first_below(long_min), b1 = std::max(b0, first_below(17)), b2 = std::max(b1, first_below(5)),
                     b3 = std::max(b2, first_below(1));
      G.vs_v[0] = b0; G.vs_v[1] = b1; G.vs_v[2] = b2; G.vs_v[3] = b3;
      G.vs_v9 = std::min(b2, std::max(b1, first_below(9)));
      G.nr_big_rows = first_below(64 * mgx::NR_BIG_UNITS + 1);       // rows of more than NR_BIG_UNITS units (mgx/nreduce.hpp)
      G.vs_edges = (unsigned)(h[b3] - h[b0]);
      G.vs_dummy = (unsigned)m + 4u;
      G.vs_long_min = long_min;
    }
  }
build_cold_lists(g);
  // (the sources' shapes -- mgx/src_shapes.hpp -- are resolved per call since round 6; what the cache held belongs to the old layout's threshold)
  G.src_shape_cache.clear();
  G.src_shapes_enabled = true;
  MGX_CATCH
}
extern "C" int mgx_csc_build_device(const int* ro, const int* ci, const float* w, int n, long long m, int* co, int* ri, float* rv,
                                    hipStream_t stream);   // mgx_layout.hip
int mgx_graph_build_csc(mgx_graph_t g) {
  MGX_TRY
  MGX_REQUIRE(g, "graph is NULL");
  use_device(g->c);
  standard_context_t& ctx = *g->c->ctx;
  graph_device_t& G = *g->g;
  const size_t n = (size_t)G.num_nodes, m = (size_t)G.num_edges;
  ctx.synchronize();
  mem_t<int> co(n + 1, ctx), ri(m, ctx);
  mem_t<float> rv(m, ctx);
  const int rc = mgx_csc_build_device(G.d_row_offsets.data(), G.d_col_indices.data(), G.d_col_values.data(), (int)n, (long long)m,
                                      co.data(), ri.data(), rv.data(), ctx.stream());
  if (rc != 0) throw mgx::mgx_error(MGX_E_HIP, std::string("mgx_graph_build_csc: ") + hipGetErrorString((hipError_t)rc));
  G.d_col_offsets = std::move(co);
  G.d_row_indices = std::move(ri);
  G.d_row_values = std::move(rv);
  G.csc_is_csr = false;
  MGX_CATCH
}
int mgx_graph_csc_read(mgx_graph_t g, int* h_col_offsets, int* h_row_indices, float* h_row_values) {
  MGX_TRY
  MGX_REQUIRE(g, "graph is NULL");
  use_device(g->c);
  g->c->ctx->synchronize();
  graph_device_t& G = *g->g;
  const size_t n = (size_t)G.num_nodes, m = (size_t)G.num_edges;
  if (h_col_offsets) MGX_HIP(mgx::dtoh(h_col_offsets, G.d_col_offsets.data(), n + 1));
  if (h_row_indices && m) MGX_HIP(mgx::dtoh(h_row_indices, G.d_row_indices.data(), m));
  if (h_row_values && m) MGX_HIP(mgx::dtoh(h_row_values, G.d_row_values.data(), m));
  MGX_CATCH
}
int mgx_graph_layout_info(mgx_graph_t g, int64_t* out8) {
  MGX_TRY
  MGX_REQUIRE(g && out8, "NULL argument");
  const graph_device_t& G = *g->g;
  for (int i = 0; i < 8; ++i) out8[i] = 0;
  if (!G.has_layout) return MGX_OK;
  out8[0] = 1;
  out8[1] = (int64_t)G.ub_units;
  out8[2] = G.d_ub_col24.size() ? 1 : 0;
  out8[3] = (int64_t)G.cold_pairs;
  out8[4] = (int64_t)G.cold_slices;
  out8[5] = (int64_t)G.ubh_units;
  out8[6] = G.cold_majority ? 1 : 0;
  auto bytes = [](auto& m) -> int64_t { return m.owned() ? (int64_t)(m.size() * sizeof(*m.data())) : 0; };
  out8[7] = bytes(G.d_layout_row_offsets) + bytes(G.d_layout_col_indices) + bytes(G.d_layout_col_values) + bytes(G.d_new_of_old) + bytes(G.d_old_of_new) +
            bytes(G.d_ub_col) + bytes(G.d_ub_col24) + bytes(G.d_ub_owner) + bytes(G.d_ubh_col24) + bytes(G.d_ubh_owner) + bytes(G.d_ub_w) + bytes(G.d_ub_w16) +
            bytes(G.d_ub_cnt) + bytes(G.d_ub_first) + bytes(G.d_cold_owner) + bytes(G.d_cold_dst) + bytes(G.d_cold_pk) + bytes(G.d_cold_cbase) +
            bytes(G.d_colds_owner) + bytes(G.d_colds_dst) + bytes(G.d_nrs_mu) + bytes(G.d_nrs_off) + bytes(G.d_nr_pos);
  MGX_CATCH
}
int mgx_graph_nr_slices_info(mgx_graph_t g, int64_t* out5) {
  MGX_TRY
  MGX_REQUIRE(g && out5, "NULL argument");
  const graph_device_t& G = *g->g;
  out5[0] = (int64_t)G.nrs_units;
  out5[1] = (int64_t)G.nrs_slices;
  out5[2] = (int64_t)G.nrs_rows;
  out5[3] = (int64_t)G.nrs_tier[2];
  out5[4] = G.nrs_units > 0 ? (int64_t)(G.nrs_first[G.nrs_slices + 1] - G.nrs_first[G.nrs_slices]) : 0;
  MGX_CATCH
}
int mgx_graph_layout_read(mgx_graph_t g, int* h_row_offsets, int* h_col_indices, int* h_new_of_old, int* h_old_of_new,
                          float* h_weights) {
  MGX_TRY
  MGX_REQUIRE(g, "graph is NULL");
  MGX_REQUIRE(g->g->has_layout, "mgx_graph_layout_read: the graph has no layout");
  use_device(g->c);
  g->c->ctx->synchronize();
  graph_device_t& G = *g->g;
  const size_t n = (size_t)G.num_nodes, m = (size_t)G.num_edges;
  if (h_row_offsets) MGX_HIP(mgx::dtoh(h_row_offsets, G.d_layout_row_offsets.data(), n + 1));
  if (h_col_indices && m) MGX_HIP(mgx::dtoh(h_col_indices, G.d_layout_col_indices.data(), m));
  if (h_new_of_old) MGX_HIP(mgx::dtoh(h_new_of_old, G.d_new_of_old.data(), n));
  if (h_old_of_new) MGX_HIP(mgx::dtoh(h_old_of_new, G.d_old_of_new.data(), n));
  if (h_weights && m) {
    MGX_REQUIRE(G.has_layout_weights, "mgx_graph_layout_read: the layout carries no weights");
    MGX_HIP(mgx::dtoh(h_weights, G.d_layout_col_values.data(), m));
  }
  MGX_CATCH
}
int mgx_graph_free(mgx_graph_t g) {
  MGX_TRY
  if (g) { use_device(g->c); delete g; }
  MGX_CATCH
}
int mgx_graph_dims(mgx_graph_t g, int* n, int64_t* m) {
  MGX_TRY
  MGX_REQUIRE(g, "graph is NULL");
  if (n) *n = g->g->num_nodes;
  if (m) *m = g->g->num_edges;
  MGX_CATCH
}

int mgx_load_mtx(const char* path, int undir, int random_w, int* n, int64_t* m, int** ro, int** ci, float** w) {
  MGX_TRY
  MGX_REQUIRE(path && n && m && ro && ci && w, "mgx_load_mtx: NULL argument");
  auto g = load_graph(path, undir != 0, random_w != 0);
  MGX_REQUIRE(g != nullptr, std::string("mgx_load_mtx: cannot read ") + path);
  *n = g->num_nodes;
  *m = g->num_edges;
  *ro = (int*)malloc(((size_t)g->num_nodes + 1) * sizeof(int));
  *ci = (int*)malloc(((size_t)g->num_edges + 1) * sizeof(int));
  *w = (float*)malloc(((size_t)g->num_edges + 1) * sizeof(float));
  memcpy(*ro, g->csr->offsets.data(), ((size_t)g->num_nodes + 1) * sizeof(int));
  memcpy(*ci, g->csr->indices.data(), (size_t)g->num_edges * sizeof(int));
  memcpy(*w, g->csr->edge_weights.data(), (size_t)g->num_edges * sizeof(float));
  MGX_CATCH
}
int mgx_load_mtx_csc(const char* path, int undir, int random_w, int genuine_csc, int* n, int64_t* m, int** ro, int** ci,
                     float** w, int** co, int** ri, float** rw) {
  MGX_TRY
  MGX_REQUIRE(path && n && m && ro && ci && w && co && ri && rw, "mgx_load_mtx_csc: NULL argument");
  auto g = load_graph(path, undir != 0, random_w != 0, genuine_csc != 0);
  MGX_REQUIRE(g != nullptr, std::string("mgx_load_mtx_csc: cannot read ") + path);
  *n = g->num_nodes;
  *m = g->num_edges;
  const size_t N = (size_t)g->num_nodes, M = (size_t)g->num_edges;
  auto dup_i = [](const std::vector<int>& v, size_t cnt) { int* p = (int*)malloc((cnt + 1) * sizeof(int)); memcpy(p, v.data(), cnt * sizeof(int)); return p; };
  auto dup_f = [](const std::vector<float>& v, size_t cnt) { float* p = (float*)malloc((cnt + 1) * sizeof(float)); memcpy(p, v.data(), cnt * sizeof(float)); return p; };
  *ro = dup_i(g->csr->offsets, N + 1); *ci = dup_i(g->csr->indices, M); *w = dup_f(g->csr->edge_weights, M);
  *co = dup_i(g->csc->offsets, N + 1); *ri = dup_i(g->csc->indices, M); *rw = dup_f(g->csc->edge_weights, M);
  MGX_CATCH
}
// binary CSR cache (include/gunrock/graph.hxx: save_graph_cache / load_graph_cache)
int mgx_graph_save_csr(const char* path, int num_nodes, int64_t num_edges, int undirected, const int* ro, const int* ci,
                       const float* w, const int* co, const int* ri, const float* rw) {
  MGX_TRY
  MGX_REQUIRE(path && ro && (ci || num_edges == 0) && num_nodes >= 0 && num_edges >= 0 && num_edges <= 2147483647LL,
              "mgx_graph_save_csr: bad argument");
  MGX_REQUIRE((co == nullptr) == (ri == nullptr), "mgx_graph_save_csr: col_offsets and row_indices go together");
  graph_t g;
  g.num_nodes = num_nodes; g.num_edges = (int)num_edges; g.undirected = undirected != 0;
  auto mk = [&](const int* o, const int* i, const float* v) {
    auto c = std::make_shared<csr_t>();
    c->num_nodes = num_nodes; c->num_edges = (int)num_edges;
    c->offsets.assign(o, o + num_nodes + 1);
    c->indices.assign(i, i + num_edges);
    if (v) c->edge_weights.assign(v, v + num_edges); else c->edge_weights.assign((size_t)num_edges, 1.0f);
    return c;
  };
  g.csr = mk(ro, ci, w);
  g.csc = co ? mk(co, ri, rw) : g.csr;
  MGX_REQUIRE(save_graph_cache(path, g), std::string("mgx_graph_save_csr: cannot write ") + path);
  MGX_CATCH
}
int mgx_graph_load_csr(const char* path, int* n, int64_t* m, int* undirected, int** ro, int** ci, float** w, int** co, int** ri,
                       float** rw) {
  MGX_TRY
  MGX_REQUIRE(path && n && m && ro && ci && w, "mgx_graph_load_csr: NULL argument");
  auto g = load_graph_cache(path);
  MGX_REQUIRE(g != nullptr, std::string("mgx_graph_load_csr: not a valid CSR cache (missing, truncated, or checksum mismatch): ") + path);
  *n = g->num_nodes; *m = g->num_edges;
  if (undirected) *undirected = g->undirected ? 1 : 0;
  const size_t N = (size_t)g->num_nodes, M = (size_t)g->num_edges;
  auto dup_i = [](const std::vector<int>& v, size_t cnt) { int* p = (int*)malloc((cnt + 1) * sizeof(int)); memcpy(p, v.data(), cnt * sizeof(int)); return p; };
  auto dup_f = [](const std::vector<float>& v, size_t cnt) { float* p = (float*)malloc((cnt + 1) * sizeof(float)); memcpy(p, v.data(), cnt * sizeof(float)); return p; };
  *ro = dup_i(g->csr->offsets, N + 1); *ci = dup_i(g->csr->indices, M); *w = dup_f(g->csr->edge_weights, M);
  const bool has_csc = g->csc != g->csr;
  if (co) *co = has_csc ? dup_i(g->csc->offsets, N + 1) : nullptr;
  if (ri) *ri = has_csc ? dup_i(g->csc->indices, M) : nullptr;
  if (rw) *rw = has_csc ? dup_f(g->csc->edge_weights, M) : nullptr;
  MGX_CATCH
}
void mgx_host_free(void* p) { free(p); }

// ---- frontier ------------------------------------------------------------------------------
int mgx_frontier_create(mgx_ctx_t c, int64_t capacity, mgx_frontier_t* out) {
  MGX_TRY
  MGX_REQUIRE(c && out && capacity >= 0, "mgx_frontier_create: bad argument");
  use_device(c);
  auto* h = new mgx_frontier_s();
  h->c = c;
  h->f = std::make_shared<frontier_t<int>>(*c->ctx, (size_t)capacity);
  *out = h;
  MGX_CATCH
}
int mgx_frontier_free(mgx_frontier_t f) {
  MGX_TRY
  if (f) { use_device(f->c); delete f; }
  MGX_CATCH
}
int mgx_frontier_load(mgx_frontier_t f, const int* host, int64_t n) {
  MGX_TRY
  MGX_REQUIRE(f && (host || n == 0) && n >= 0, "mgx_frontier_load: bad argument");
  use_device(f->c);
  f->f->resize((size_t)n);   // throws MGX_E_FRONTIER_OVERFLOW
  mgx::frontier_touched();
  MGX_HIP(mgx::htod(f->f->data()->data(), host, (size_t)n));
  MGX_CATCH
}
int mgx_frontier_fill_iota(mgx_frontier_t f, int64_t n) {
  MGX_TRY
  MGX_REQUIRE(f && n >= 0, "mgx_frontier_fill_iota: bad argument");
  use_device(f->c);
  f->f->resize((size_t)n);
  mgx::frontier_touched();
  int* p = f->f->data()->data();
  mgx::transform([=] __device__(int i) { p[i] = i; }, n, *f->c->ctx);
  MGX_CATCH
}
int mgx_frontier_fill(mgx_frontier_t f, int value, int64_t n) {
  MGX_TRY
  MGX_REQUIRE(f && n >= 0, "mgx_frontier_fill: bad argument");
  use_device(f->c);
  f->f->resize((size_t)n);
  mgx::frontier_touched();
  int* p = f->f->data()->data();
  mgx::transform([=] __device__(int i) { p[i] = value; }, n, *f->c->ctx);
  MGX_CATCH
}
int mgx_frontier_read(mgx_frontier_t f, int* host, int64_t cap, int64_t* n) {
  MGX_TRY
  MGX_REQUIRE(f && n, "mgx_frontier_read: bad argument");
  use_device(f->c);
  f->c->ctx->synchronize();
  *n = (int64_t)f->f->size();
  if (host) {
    MGX_REQUIRE(cap >= *n, "mgx_frontier_read: host buffer too small");
    MGX_HIP(mgx::dtoh(host, f->f->data()->data(), f->f->size()));
  }
  MGX_CATCH
}
int mgx_frontier_resize(mgx_frontier_t f, int64_t n) {
  MGX_TRY
  MGX_REQUIRE(f && n >= 0, "mgx_frontier_resize: bad argument");
  f->f->resize((size_t)n);
  MGX_CATCH
}
int mgx_frontier_size(mgx_frontier_t f, int64_t* n) {
  MGX_TRY
  MGX_REQUIRE(f && n, "NULL argument");
  *n = (int64_t)f->f->size();
  MGX_CATCH
}
int mgx_frontier_capacity(mgx_frontier_t f, int64_t* n) {
  MGX_TRY
  MGX_REQUIRE(f && n, "NULL argument");
  *n = (int64_t)f->f->capacity();
  MGX_CATCH
}
int mgx_frontier_device_ptr(mgx_frontier_t f, int** p) {
  MGX_TRY
  MGX_REQUIRE(f && p, "NULL argument");
  *p = f->f->data()->data();
  f->f->mark_exposed();
  MGX_CATCH
}

// ---- building blocks -----------------------------------------------------------------------
int mgx_scan_exclusive_i32(mgx_ctx_t c, const int* d_in, int64_t n, int* d_out, int64_t* total) {
  MGX_TRY
  MGX_REQUIRE(c && (d_in || n == 0) && (d_out || n == 0) && n >= 0, "mgx_scan_exclusive_i32: bad argument");
  use_device(c);
  c->ctx->reserve_scratch(mgx::scan_scratch_bytes(n));
  long long t = 0;
  mgx::transform_scan([=] __device__(long long i) { return d_in[i]; }, n, d_out, *c->ctx, &t);
  if (total) *total = t;
  MGX_CATCH
}

int mgx_scan_frontier_degrees(mgx_graph_t g, mgx_frontier_t in, int use_csc, int64_t* total) {
  MGX_TRY
  MGX_REQUIRE(g && in, "NULL argument");
  use_device(g->c);
  standard_context_t& ctx = *g->c->ctx;
  g->g->ensure_scanned(in->f->capacity(), ctx);
  const int* input_data = in->f->data()->data();
  const int* offsets = use_csc ? g->g->d_col_offsets.data() : g->g->d_row_offsets.data();
  long long t = 0;
  mgx::transform_scan(
      [=] __device__(long long i) {
        const int v = input_data[i];
        return offsets[v + 1] - offsets[v];
      },
      (long long)in->f->size(), g->g->d_scanned_row_offsets.data(), ctx, &t);
  if (total) *total = t;
  MGX_CATCH
}

int mgx_lbs_expand_debug(mgx_graph_t g, mgx_frontier_t in, int64_t total, int* host_seg, int* host_rank) {
  MGX_TRY
  MGX_REQUIRE(g && in && total >= 0 && (total == 0 || (host_seg && host_rank)), "bad argument");
  use_device(g->c);
  standard_context_t& ctx = *g->c->ctx;
  if (total == 0) return MGX_OK;
  mem_t<int> seg((size_t)total, ctx), rank((size_t)total, ctx);
  int* ps = seg.data();
  int* pr = rank.data();
  mgx::transform_lbs([=] __device__(int idx, int s, int r) { ps[idx] = s; pr[idx] = r; }, total,
                     g->g->d_scanned_row_offsets.data(), (long long)in->f->size(), ctx);
  ctx.synchronize();
  MGX_HIP(mgx::dtoh(host_seg, ps, (size_t)total));
  MGX_HIP(mgx::dtoh(host_rank, pr, (size_t)total));
  MGX_CATCH
}

int mgx_compact_i32(mgx_ctx_t c, const int* d_in, int64_t n, int drop_value, int* d_out, int64_t* kept) {
  MGX_TRY
  MGX_REQUIRE(c && (d_in || n == 0) && n >= 0, "mgx_compact_i32: bad argument");
  use_device(c);
  c->ctx->reserve_scratch(mgx::scan_scratch_bytes(n));
  auto compact = mgx::transform_compact(n, *c->ctx);
  const long long k = compact.upsweep([=] __device__(long long i) { return d_in[i] != drop_value; });
  compact.downsweep([=] __device__(long long d, long long s) { d_out[d] = d_in[s]; });
  c->ctx->synchronize();
  if (kept) *kept = k;
  MGX_CATCH
}

int mgx_segreduce_f32_plus(mgx_graph_t g, mgx_frontier_t in, int push, const float* v, float id, float* red,
                           int64_t* nz) {
  return segreduce_impl<float, mgx::plus_t<float>>(g, in, push, v, id, red, nz);
}
int mgx_segreduce_i32_min(mgx_graph_t g, mgx_frontier_t in, int push, const int* v, int id, int* red, int64_t* nz) {
  return segreduce_impl<int, mgx::minimum_t<int>>(g, in, push, v, id, red, nz);
}
int mgx_segreduce_i32_max(mgx_graph_t g, mgx_frontier_t in, int push, const int* v, int id, int* red, int64_t* nz) {
  return segreduce_impl<int, mgx::maximum_t<int>>(g, in, push, v, id, red, nz);
}

// ---- BFS -----------------------------------------------------------------------------------
static bfs::bfs_enactor_t& bfs_enactor(mgx_bfs_t p) {
  if (!p->e) p->e.reset(new bfs::bfs_enactor_t(*p->g->c->ctx, p->g->g->num_nodes, p->g->g->num_edges));
  return *p->e;
}

int mgx_bfs_create(mgx_graph_t g, int src, mgx_bfs_t* out) {
  MGX_TRY
  MGX_REQUIRE(g && out, "NULL argument");
  MGX_REQUIRE(src >= 0 && src < g->g->num_nodes, "mgx_bfs_create: src out of range");
  use_device(g->c);
  auto* h = new mgx_bfs_s();
  h->g = g;
  h->p = std::make_shared<bfs::bfs_problem_t>(g->g, (size_t)src, *g->c->ctx);
  *out = h;
  MGX_CATCH
}
int mgx_bfs_reset(mgx_bfs_t p, int src) {
  MGX_TRY
  MGX_REQUIRE(p, "NULL argument");
  MGX_REQUIRE(src >= 0 && src < p->g->g->num_nodes, "mgx_bfs_reset: src out of range");
  use_device(p->g->c);
  p->p->reset((size_t)src, *p->g->c->ctx);
  if (p->visited_mask.size()) {        // the idempotent mode's bitmask: only the source seen
    unsigned* const mask = p->visited_mask.data();
    MGX_HIP(hipMemsetAsync(mask, 0, p->visited_mask.size() * sizeof(unsigned), p->g->c->ctx->stream()));
    mgx::transform([=] __device__(int) { mask[src >> 5] = 1u << (src & 31); }, 1, *p->g->c->ctx);
  }
  MGX_CATCH
}
int mgx_bfs_free(mgx_bfs_t p) {
  MGX_TRY
  if (p) { use_device(p->g->c); delete p; }
  MGX_CATCH
}
int mgx_bfs_labels(mgx_bfs_t p, int* host) {
  MGX_TRY
  MGX_REQUIRE(p && host, "NULL argument");
  use_device(p->g->c);
  p->g->c->ctx->synchronize();
  MGX_HIP(mgx::dtoh(host, p->p->d_labels.data(), (size_t)p->g->g->num_nodes));
  MGX_CATCH
}
int mgx_bfs_preds(mgx_bfs_t p, int* host) {
  MGX_TRY
  MGX_REQUIRE(p && host, "NULL argument");
  use_device(p->g->c);
  p->g->c->ctx->synchronize();
  MGX_HIP(mgx::dtoh(host, p->p->d_preds.data(), (size_t)p->g->g->num_nodes));
  MGX_CATCH
}
int mgx_bfs_labels_device(mgx_bfs_t p, int** d) {
  MGX_TRY
  MGX_REQUIRE(p && d, "NULL argument");
  *d = p->p->d_labels.data();
  MGX_CATCH
}

int mgx_bfs_advance(mgx_bfs_t p, mgx_frontier_t in, mgx_frontier_t out, int iteration, int64_t* front) {
  MGX_TRY
  MGX_REQUIRE(p && in && out, "NULL argument");
  use_device(p->g->c);
  const int r = oprtr::advance::advance_forward_kernel<bfs::bfs_problem_t, bfs::bfs_functor_t, false, true>(
      p->p, in->f, out->f, iteration, *p->g->c->ctx);
  if (front) *front = r;
  MGX_CATCH
}
int mgx_bfs_filter(mgx_bfs_t p, mgx_frontier_t in, mgx_frontier_t out, int iteration, int64_t* kept) {
  MGX_TRY
  MGX_REQUIRE(p && in && out, "NULL argument");
  use_device(p->g->c);
  const int r = oprtr::filter::filter_kernel<bfs::bfs_problem_t, bfs::bfs_functor_t>(p->p, in->f, out->f, iteration,
                                                                                   *p->g->c->ctx);
  if (kept) *kept = r;
  MGX_CATCH
}
int mgx_bfs_advance_filter_fused(mgx_bfs_t p, mgx_frontier_t in, mgx_frontier_t out, int iteration, int64_t* kept) {
  MGX_TRY
  MGX_REQUIRE(p && in && out, "NULL argument");
  use_device(p->g->c);
  const int r = oprtr::advance::advance_filter_fused_kernel<bfs::bfs_problem_t, bfs::bfs_functor_t>(
      p->p, in->f, out->f, iteration, *p->g->c->ctx);
  if (kept) *kept = r;
  MGX_CATCH
}
int mgx_bfs_gen_unvisited(mgx_bfs_t p, mgx_frontier_t indices, mgx_frontier_t unvisited, int iteration,
                          int64_t* kept) {
  MGX_TRY
  MGX_REQUIRE(p && indices && unvisited, "NULL argument");
  use_device(p->g->c);
  const int r = oprtr::advance::gen_unvisited_kernel<bfs::bfs_problem_t, bfs::bfs_functor_t>(
      p->p, indices->f, unvisited->f, iteration, *p->g->c->ctx);
  if (kept) *kept = r;
  MGX_CATCH
}
int mgx_bfs_sparse_to_dense(mgx_bfs_t p, mgx_frontier_t sparse, mgx_frontier_t dense, int iteration) {
  MGX_TRY
  MGX_REQUIRE(p && sparse && dense, "NULL argument");
  use_device(p->g->c);
  oprtr::advance::sparse_to_dense_kernel<bfs::bfs_problem_t, bfs::bfs_functor_t>(p->p, sparse->f, dense->f,
                                                                                iteration, *p->g->c->ctx);
  MGX_CATCH
}
int mgx_bfs_advance_backward(mgx_bfs_t p, mgx_frontier_t unvisited, mgx_frontier_t bitmap, mgx_frontier_t bitmap_out,
                             int iteration, int64_t* front) {
  MGX_TRY
  MGX_REQUIRE(p && unvisited && bitmap && bitmap_out, "NULL argument");
  use_device(p->g->c);
  const int r = oprtr::advance::advance_backward_kernel<bfs::bfs_problem_t, bfs::bfs_functor_t>(
      p->p, unvisited->f, bitmap->f, bitmap_out->f, iteration, *p->g->c->ctx);
  if (front) *front = r;
  MGX_CATCH
}
int mgx_bfs_enact_pushpull(mgx_bfs_t p, float threshold, int64_t* stats) {
  MGX_TRY
  MGX_REQUIRE(p, "NULL argument");
  use_device(p->g->c);
  bfs::bfs_enactor_t& e = bfs_enactor(p);
  e.enact_pushpull(p->p, threshold, *p->g->c->ctx);
  p->g->c->ctx->synchronize();
  if (stats) {
    stats[0] = e.pushed_iterations;
    stats[1] = e.total_iterations;
    stats[2] = e.pushed_edges;
    stats[3] = e.pulled_edges;
  }
  MGX_CATCH
}

static mem_t<unsigned>& bfs_mask(mgx_bfs_t p) {
  if (!p->visited_mask.size()) {
    const size_t words = ((size_t)p->g->g->num_nodes + 31) / 32 + 1;
    p->visited_mask = mem_t<unsigned>(words, *p->g->c->ctx);
    unsigned* const mask = p->visited_mask.data();
    const int src = p->p->src;           // the state mgx_bfs_reset leaves: only the source seen
    MGX_HIP(hipMemsetAsync(mask, 0, words * sizeof(unsigned), p->g->c->ctx->stream()));
    mgx::transform([=] __device__(int) { mask[src >> 5] = 1u << (src & 31); }, 1, *p->g->c->ctx);
  }
  return p->visited_mask;
}
int mgx_bfs_advance_idempotent(mgx_bfs_t p, mgx_frontier_t in, mgx_frontier_t out, int iteration, int64_t* front) {
  MGX_TRY
  MGX_REQUIRE(p && in && out, "NULL argument");
  use_device(p->g->c);
  const int r = oprtr::advance::advance_forward_kernel<bfs::bfs_problem_t, bfs::bfs_idempotent_functor_t, true, true>(
      p->p, in->f, out->f, iteration, *p->g->c->ctx);
  if (front) *front = r;
  MGX_CATCH
}
int mgx_bfs_uniquify(mgx_bfs_t p, mgx_frontier_t in, mgx_frontier_t out, int iteration, int64_t* kept) {
  MGX_TRY
  MGX_REQUIRE(p && in && out, "NULL argument");
  use_device(p->g->c);
  mem_t<unsigned>& mask = bfs_mask(p);
  oprtr::filter::uniquify_kernel<bfs::bfs_problem_t, bfs::bfs_idempotent_functor_t>(p->p, (unsigned char*)mask.data(), in->f,
                                                                                    out->f, iteration, *p->g->c->ctx);
  if (kept) *kept = (int64_t)out->f->size();
  MGX_CATCH
}
int mgx_bfs_enact_idempotent(mgx_bfs_t p, int64_t* stats) {
  MGX_TRY
  MGX_REQUIRE(p, "NULL argument");
  use_device(p->g->c);
  bfs::bfs_enactor_t& e = bfs_enactor(p);
  e.enact_idempotent(p->p, bfs_mask(p), *p->g->c->ctx);
  p->g->c->ctx->synchronize();
  if (stats) {
    stats[0] = e.total_iterations;
    stats[1] = e.idempotent_edges;
  }
  MGX_CATCH
}

static void fill_bfs_stats(int64_t* out24, const bfs::bfs_run_stats_t& L) {
  out24[0] = L.levels;
  out24[1] = L.reached;
  out24[2] = L.m_t;
  out24[3] = L.push_edges;
  out24[4] = L.pull_edges;
  out24[5] = L.push_levels;
  out24[6] = L.kernel_launches;
  out24[7] = L.kernel_ns;
  out24[8] = L.frontier_vertices;
  out24[9] = L.claims;
  const bfs::bfs_kernel_stats_t& D = L.dominant ? L.stream : L.wave;
  out24[10] = D.launches;
  out24[11] = D.ns;
  out24[12] = D.edges;
  out24[13] = D.vertices;
  out24[14] = L.dominant;
  out24[15] = L.small_levels;
  out24[16] = L.slots;
  out24[17] = L.dense_slots;
  out24[18] = L.vshort_slots;
  out24[19] = L.lazy_slots;
  out24[20] = L.cold_slots;
  out24[21] = L.mini_slots;
}
int mgx_bfs_run(mgx_bfs_t p, int src, int mode, float alpha, int64_t* stats) { return mgx_bfs_run_stats(p, src, mode, alpha, stats, 16); }
int mgx_bfs_run_stats(mgx_bfs_t p, int src, int mode, float alpha, int64_t* stats, int cap) {
  MGX_TRY
  MGX_REQUIRE(p, "NULL argument");
  MGX_REQUIRE(cap >= 0, "mgx_bfs_run_stats: negative capacity");
  MGX_REQUIRE(src >= 0 && src < p->g->g->num_nodes, "mgx_bfs_run: src out of range");
  MGX_REQUIRE(mode == MGX_BFS_PUSH || mode == MGX_BFS_DIRECTION_OPT, "mgx_bfs_run: unknown mode");
  use_device(p->g->c);
  standard_context_t& ctx = *p->g->c->ctx;
  if (!p->fe) p->fe.reset(new bfs::bfs_fused_enactor_t(ctx, p->g->g->num_nodes));
  if (p->time_kernels >= 0) p->fe->fused->time_kernels = p->time_kernels;
  p->p->src = src;
  p->fe->enact(p->p, ctx, mode == MGX_BFS_DIRECTION_OPT, alpha);
  fill_bfs_stats(p->last_stats, p->fe->last);
  constexpr int have = (int)(sizeof(p->last_stats) / sizeof(p->last_stats[0]));
  if (stats) memcpy(stats, p->last_stats, sizeof(int64_t) * (size_t)(cap < have ? cap : have));
  MGX_CATCH
}
int mgx_bfs_run_many(mgx_bfs_t p, const int* sources, int count, int mode, float alpha, int64_t* stats, int cap, int* reruns) {
  MGX_TRY
  MGX_REQUIRE(p && (sources || count == 0), "NULL argument");
  MGX_REQUIRE(count >= 0 && count <= (1 << 20), "mgx_bfs_run_many: count out of range");
  MGX_REQUIRE(cap >= 0, "mgx_bfs_run_many: negative capacity");
  MGX_REQUIRE(mode == MGX_BFS_PUSH || mode == MGX_BFS_DIRECTION_OPT, "mgx_bfs_run_many: unknown mode");
  for (int i = 0; i < count; ++i) MGX_REQUIRE(sources[i] >= 0 && sources[i] < p->g->g->num_nodes, "mgx_bfs_run_many: src out of range");
  if (reruns) *reruns = 0;
  if (count == 0) return MGX_OK;
  use_device(p->g->c);
  standard_context_t& ctx = *p->g->c->ctx;
  if (!p->fe) p->fe.reset(new bfs::bfs_fused_enactor_t(ctx, p->g->g->num_nodes));
  p->fe->fused->time_kernels = 0;                       // (per-launch events belong to mgx_bfs_run)
  std::vector<bfs::bfs_run_stats_t> all;
  const int rr = p->fe->enact_many(p->p, ctx, sources, count, all, mode == MGX_BFS_DIRECTION_OPT, alpha);
  if (reruns) *reruns = rr;
  constexpr int have = (int)(sizeof(p->last_stats) / sizeof(p->last_stats[0]));
  const int take = cap < have ? cap : have;
  for (int i = 0; i < count; ++i) {
    fill_bfs_stats(p->last_stats, all[(size_t)i]);
    if (stats && take > 0) memcpy(stats + (size_t)i * (size_t)cap, p->last_stats, sizeof(int64_t) * (size_t)take);
  }
  MGX_CATCH
}
int mgx_bfs_level_trace(mgx_bfs_t p, int cap, int64_t* level_nf, int64_t* level_edges, int* levels) {
  MGX_TRY
  MGX_REQUIRE(p && levels, "NULL argument");
  MGX_REQUIRE(p->fe != nullptr, "mgx_bfs_level_trace: no mgx_bfs_run yet");
  const auto& tr = p->fe->last.trace;
  *levels = (int)tr.size();
  for (int i = 0; i < (int)tr.size() && i < cap; ++i) {
    if (level_nf) level_nf[i] = tr[i].first;
    if (level_edges) level_edges[i] = tr[i].second;
  }
  MGX_CATCH
}
int mgx_bfs_set_kernel_timing(mgx_bfs_t p, int on) {
  MGX_TRY
  MGX_REQUIRE(p, "NULL argument");
  p->time_kernels = on < 0 ? 0 : on;
  MGX_CATCH
}
int mgx_bfs_kernel_times(mgx_bfs_t p, int64_t* out8) {
  MGX_TRY
  MGX_REQUIRE(p && out8, "NULL argument");
  MGX_REQUIRE(p->fe != nullptr, "mgx_bfs_kernel_times: no mgx_bfs_run yet");
  const bfs::bfs_run_stats_t& L = p->fe->last;
  out8[0] = L.stream.launches; out8[1] = L.stream.ns; out8[2] = L.stream.edges; out8[3] = L.stream.vertices;
  out8[4] = L.wave.launches; out8[5] = L.wave.ns; out8[6] = L.wave.edges; out8[7] = L.wave.vertices;
  MGX_CATCH
}
int mgx_bfs_level_kernel_times(mgx_bfs_t p, int cap, float* stream_ms, float* wave_ms) {
  MGX_TRY
  MGX_REQUIRE(p && stream_ms && wave_ms, "NULL argument");
  MGX_REQUIRE(p->fe != nullptr, "mgx_bfs_level_kernel_times: no mgx_bfs_run yet");
  for (int i = 0; i < cap && i < 64; ++i) {
    stream_ms[i] = p->fe->fused->level_stream_ms[i];
    wave_ms[i] = p->fe->fused->level_wave_ms[i];
  }
  MGX_CATCH
}
int mgx_bfs_level_times(mgx_bfs_t p, int cap, float* ms, int* levels) {
  MGX_TRY
  MGX_REQUIRE(p && levels, "NULL argument");
  MGX_REQUIRE(p->fe != nullptr, "mgx_bfs_level_times: no mgx_bfs_run yet");
  const auto& t = p->fe->last.level_ms;
  *levels = (int)t.size();
  for (int i = 0; i < (int)t.size() && i < cap; ++i)
    if (ms) ms[i] = t[i];
  MGX_CATCH
}
int mgx_bfs_level_claims(mgx_bfs_t p, int cap, int64_t* claims) {
  MGX_TRY
  MGX_REQUIRE(p && claims, "NULL argument");
  MGX_REQUIRE(p->fe != nullptr, "mgx_bfs_level_claims: no mgx_bfs_run yet");
  for (int i = 0; i < cap && i < 64; ++i) claims[i] = p->fe->last.claims_level[i];
  MGX_CATCH
}
int mgx_bfs_batch_times(mgx_bfs_t p, int cap, float* ms, int* batches) {
  MGX_TRY
  MGX_REQUIRE(p && batches, "NULL argument");
  MGX_REQUIRE(p->fe != nullptr, "mgx_bfs_batch_times: no mgx_bfs_run yet");
  const auto& b = p->fe->last.batch_ms;
  *batches = (int)b.size();
  for (int i = 0; i < (int)b.size() && i < cap; ++i)
    if (ms) ms[i] = b[i];
  MGX_CATCH
}

// ---- vertex-range partitioned BFS (per-rank pieces; the exchange is the host's job) -----------
int mgx_dbfs_create(mgx_ctx_t c, int n_global, int ranks, int rank, int64_t m_local, const int* d_row_offsets,
                    const int* d_col_indices, int* d_bins, int64_t bin_capacity, mgx_dbfs_t* out) {
  MGX_TRY
  MGX_REQUIRE(c && out && d_row_offsets && (d_col_indices || m_local == 0), "mgx_dbfs_create: NULL argument");
  MGX_REQUIRE(n_global > 0 && ranks >= 1 && ranks <= mgx::DBFS_MAX_RANKS && rank >= 0 && rank < ranks,
              "mgx_dbfs_create: bad partition");
  use_device(c);
  const int chunk = (n_global + ranks - 1) / ranks;
  const int lo = rank * chunk < n_global ? rank * chunk : n_global;
  const int hi = (rank + 1) * chunk < n_global ? (rank + 1) * chunk : n_global;
  auto* h = new mgx_dbfs_s();
  h->c = c;
  MGX_REQUIRE(d_bins == nullptr || bin_capacity >= chunk, "mgx_dbfs_create: bin capacity must be >= ceil(n/ranks)");
  h->st.init(*c->ctx, n_global, lo, hi, ranks, rank, d_row_offsets, d_col_indices, m_local, d_bins, bin_capacity);
  *out = h;
  MGX_CATCH
}
int mgx_dbfs_free(mgx_dbfs_t h) {
  MGX_TRY
  if (h) { use_device(h->c); delete h; }
  MGX_CATCH
}
int mgx_dbfs_range(mgx_dbfs_t h, int* lo, int* hi) {
  MGX_TRY
  MGX_REQUIRE(h, "NULL argument");
  if (lo) *lo = h->st.v_lo;
  if (hi) *hi = h->st.v_hi;
  MGX_CATCH
}
int mgx_dbfs_reset(mgx_dbfs_t h, int src_global) {
  MGX_TRY
  MGX_REQUIRE(h && src_global >= 0 && src_global < h->st.n_global, "mgx_dbfs_reset: bad argument");
  use_device(h->c);
  mgx::dbfs_reset(h->st, src_global, *h->c->ctx);
  MGX_CATCH
}
int mgx_dbfs_expand(mgx_dbfs_t h, int64_t* counts, int64_t* edges) {
  MGX_TRY
  MGX_REQUIRE(h && counts, "NULL argument");
  use_device(h->c);
  long long e = 0;
  mgx::dbfs_expand(h->st, *h->c->ctx, &e);
  for (int r = 0; r < h->st.ranks; ++r) {
    counts[r] = (int64_t)h->st.host_counters[r];
    if (counts[r] > h->st.bin_cap) throw mgx::mgx_error(MGX_E_FRONTIER_OVERFLOW, "mgx_dbfs_expand: send bin overflow");
  }
  if (edges) *edges = e;
  MGX_CATCH
}
int mgx_dbfs_bins(mgx_dbfs_t h, int** d_bins, int64_t* bin_cap) {
  MGX_TRY
  MGX_REQUIRE(h && d_bins && bin_cap, "NULL argument");
  *d_bins = h->st.bins.data();
  *bin_cap = h->st.bin_cap;
  MGX_CATCH
}
int mgx_dbfs_receive(mgx_dbfs_t h, const int* d_ids, int64_t count, int label) {
  MGX_TRY
  MGX_REQUIRE(h && (d_ids || count == 0) && count >= 0, "bad argument");
  use_device(h->c);
  mgx::dbfs_receive(h->st, d_ids, count, label, *h->c->ctx);
  MGX_CATCH
}
int mgx_dbfs_swap(mgx_dbfs_t h, int64_t* frontier_size) {
  MGX_TRY
  MGX_REQUIRE(h, "NULL argument");
  use_device(h->c);
  const long long f = mgx::dbfs_swap(h->st, *h->c->ctx);
  if (frontier_size) *frontier_size = f;
  MGX_CATCH
}
int mgx_dbfs_labels(mgx_dbfs_t h, int* host_labels) {
  MGX_TRY
  MGX_REQUIRE(h && host_labels, "NULL argument");
  use_device(h->c);
  h->c->ctx->synchronize();
  MGX_HIP(mgx::dtoh(host_labels, h->st.labels.data(), (size_t)h->st.n_local));
  MGX_CATCH
}

// ---- vertex-range partitioned SSSP (per-rank pieces; mgx/sssp_dist.hpp) -----------------------------------------
int mgx_dsssp_create(mgx_ctx_t c, int n_global, int ranks, int rank, int64_t m_local, const int* d_row_offsets,
                     const int* d_col_indices, const float* d_weights, mgx_dsssp_t* out) {
  MGX_TRY
  MGX_REQUIRE(c && out && d_row_offsets && ((d_col_indices && d_weights) || m_local == 0), "mgx_dsssp_create: NULL argument");
  MGX_REQUIRE(n_global > 0 && ranks >= 1 && ranks <= mgx::DBFS_MAX_RANKS && rank >= 0 && rank < ranks, "mgx_dsssp_create: bad partition");
  use_device(c);
  const int chunk = (n_global + ranks - 1) / ranks;
  const int lo = rank * chunk < n_global ? rank * chunk : n_global;
  const int hi = (rank + 1) * chunk < n_global ? (rank + 1) * chunk : n_global;
  auto* h = new mgx_dsssp_s();
  h->c = c;
  h->st.init(*c->ctx, n_global, lo, hi, ranks, rank, d_row_offsets, d_col_indices, d_weights, m_local);
  *out = h;
  MGX_CATCH
}
int mgx_dsssp_free(mgx_dsssp_t h) {
  MGX_TRY
  if (h) { use_device(h->c); delete h; }
  MGX_CATCH
}
int mgx_dsssp_reset(mgx_dsssp_t h, int src_global) {
  MGX_TRY
  MGX_REQUIRE(h && src_global >= 0 && src_global < h->st.n_global, "mgx_dsssp_reset: bad argument");
  use_device(h->c);
  mgx::dsssp_reset(h->st, src_global, *h->c->ctx);
  MGX_CATCH
}
int mgx_dsssp_expand(mgx_dsssp_t h, int64_t* counts, int64_t* edges) {
  MGX_TRY
  MGX_REQUIRE(h && counts, "NULL argument");
  use_device(h->c);
  long long e = 0;
  mgx::dsssp_expand(h->st, *h->c->ctx, &e);
  for (int r = 0; r < h->st.ranks; ++r) {
    counts[r] = (int64_t)h->st.host_counters[r];
    if (counts[r] > h->st.bin_cap) throw mgx::mgx_error(MGX_E_FRONTIER_OVERFLOW, "mgx_dsssp_expand: send bin overflow");
  }
  if (edges) *edges = e;
  MGX_CATCH
}
int mgx_dsssp_bins(mgx_dsssp_t h, uint64_t** d_bins, int64_t* bin_cap) {
  MGX_TRY
  MGX_REQUIRE(h && d_bins && bin_cap, "NULL argument");
  *d_bins = (uint64_t*)h->st.bins.data();
  *bin_cap = h->st.bin_cap;
  MGX_CATCH
}
int mgx_dsssp_receive(mgx_dsssp_t h, const uint64_t* d_pairs, int64_t count) {
  MGX_TRY
  MGX_REQUIRE(h && (d_pairs || count == 0) && count >= 0, "bad argument");
  use_device(h->c);
  mgx::dsssp_receive(h->st, (const unsigned long long*)d_pairs, count, *h->c->ctx);
  MGX_CATCH
}
int mgx_dsssp_swap(mgx_dsssp_t h, int64_t* frontier_size) {
  MGX_TRY
  MGX_REQUIRE(h, "NULL argument");
  use_device(h->c);
  const long long f = mgx::dsssp_swap(h->st, *h->c->ctx);
  if (frontier_size) *frontier_size = f;
  MGX_CATCH
}
int mgx_dsssp_distances(mgx_dsssp_t h, float* host_dist_local) {
  MGX_TRY
  MGX_REQUIRE(h && host_dist_local, "NULL argument");
  use_device(h->c);
  h->c->ctx->synchronize();
  MGX_HIP(mgx::dtoh((unsigned*)host_dist_local, h->st.dist.data(), (size_t)h->st.n_local));
  MGX_CATCH
}

int mgx_dsssp_run(mgx_dsssp_t h, mgx_comm_t comm, int src_global, int64_t* out4) {
  MGX_TRY
  MGX_REQUIRE(h && out4 && src_global >= 0 && src_global < h->st.n_global, "mgx_dsssp_run: bad argument");
  MGX_REQUIRE(comm || h->st.ranks == 1, "mgx_dsssp_run: a communicator is needed for more than one rank");
  MGX_REQUIRE(!comm || (comm->cm.ranks == h->st.ranks && comm->cm.rank == h->st.rank), "mgx_dsssp_run: communicator and engine disagree on the partition");
  use_device(h->c);
  mgx::comm_t none;
  long long o[4];
  mgx::dsssp_run(h->st, comm ? comm->cm : none, h->run_bufs, src_global, *h->c->ctx, o);
  for (int i = 0; i < 4; ++i) out4[i] = o[i];
  MGX_CATCH
}

// ---- partitioned BFS, generation 2: fused kernels per rank + bitmap exchange (mgx/bfs_dist2.hpp) ----
int mgx_dbfs2_create(mgx_ctx_t c, int n_global, int ranks, int rank, const int* d_row_offsets_local,
                     const int* d_col_indices_global, unsigned* d_newbits, mgx_dbfs2_t* out) {
  MGX_TRY
  MGX_REQUIRE(c && out && d_row_offsets_local && d_newbits, "mgx_dbfs2_create: NULL argument");
  MGX_REQUIRE(n_global > 0 && ranks >= 1 && ranks <= 64 && rank >= 0 && rank < ranks, "mgx_dbfs2_create: bad partition");
  use_device(c);
  auto* h = new mgx_dbfs2_s();
  h->c = c;
  h->st.init(*c->ctx, n_global, ranks, rank, d_row_offsets_local, d_col_indices_global, d_newbits);
  c->ctx->synchronize();
  *out = h;
  MGX_CATCH
}
int mgx_dbfs2_build_units(mgx_dbfs2_t h, int64_t* units) {
  MGX_TRY
  MGX_REQUIRE(h, "NULL argument");
  use_device(h->c);
  mgx::d2_state_t& st = h->st;
  if (units) *units = 0;
  if (st.ub_col) { if (units) *units = st.ub_units; return MGX_OK; }
  const int long_min = st.fs->long_min;
  if (long_min != 64 || st.n_local <= 0) return MGX_OK;     // (a unit is 64 entries: only with the default long-row threshold)
  h->c->ctx->synchronize();
  int *owner = nullptr, *ucol = nullptr, *ufirst = nullptr;
  unsigned char* ucnt = nullptr;
  long long U = 0, Up = 0;
  const int rc = mgx_units_build_device(st.row_offsets, st.col_indices, st.n_local, long_min, 0x7FFFFFFF, 6, 0u, &owner, &ucol, &ucnt, &ufirst, &U, &Up,
                                        h->c->ctx->stream());
  if (rc != 0) throw mgx::mgx_error(MGX_E_HIP, std::string("partitioned BFS, unit blocks: ") + hipGetErrorString((hipError_t)rc));
  if (ucnt) (void)hipFree(ucnt);
  if (ufirst) (void)hipFree(ufirst);
  if (U <= 0) return MGX_OK;
  hipLaunchKernelGGL(mgx::k_d2_owner_global, dim3((unsigned)((Up + mgx::BLOCK - 1) / mgx::BLOCK)), dim3(mgx::BLOCK), 0, h->c->ctx->stream(), owner, Up,
                     st.ranks, st.rank, st.n_local, st.n_global);
  h->c->ctx->synchronize();
  st.ub_owner = owner; st.ub_col = ucol; st.ub_units = U; st.ub_units_pad = Up;
  if (const char* e = mgx::env("MGX_DIST_DENSE_DIV")) { const int d = atoi(e); if (d >= 0) st.dense_div = (unsigned)d; }
  // the short rows vertex by vertex (MGX_DIST_VSHORT=0: never; N: when a level holds 1 / N of their edges): needs rows by
  // non-increasing degree (hub-first global ids, cyclic ownership: they are) -- checked here, on the host
  {
    int vdiv = 8;
    if (const char* e = mgx::env("MGX_DIST_VSHORT")) vdiv = atoi(e);
    standard_context_t& ctx = *h->c->ctx;
    const int n = st.n_local;
    if (n > 0) {
      // deferred hot marks only on a shard big enough to pay for their bitmaps (d2_state_t::defer_pays)
      int m_last = 0;
      MGX_HIP(mgx::dtoh(&m_last, st.row_offsets + n, 1));
      int defer_mode = 1;
      if (const char* e = mgx::env("MGX_DIST_DEFER")) defer_mode = atoi(e);
      st.defer_pays = defer_mode == 2 || (long long)m_last >= mgx::d2_state_t::D2_DEFER_MIN_ENTRIES;
    }
    if (vdiv > 0 && n > 0) {
      std::vector<int> hro((size_t)n + 1);
      MGX_HIP(mgx::dtoh(hro.data(), st.row_offsets, (size_t)n + 1));
      bool sorted = true;
      for (int i = 1; i < n && sorted; ++i) sorted = hro[i + 1] - hro[i] <= hro[i] - hro[i - 1];
      const long long m_local = hro[n];
      if (sorted && m_local > 0) {
        auto first_below = [&](int d) {            // first row with degree < d
          size_t lo = 0, hi = (size_t)n;
          while (lo < hi) { const size_t mid = (lo + hi) / 2; if (hro[mid + 1] - hro[mid] >= d) lo = mid + 1; else hi = mid; }
          return (unsigned)lo;
        };
        const unsigned b0 = first_below(long_min), b1 = std::max(b0, first_below(17)), b2 = std::max(b1, first_below(5)), b3 = std::max(b2, first_below(1));
        st.vs_v[0] = b0; st.vs_v[1] = b1; st.vs_v[2] = b2; st.vs_v[3] = b3;
        st.vs_v9 = std::min(b2, std::max(b1, first_below(9)));
        st.vs_edges = (unsigned)(hro[b3] - hro[b0]);
        st.col_pad = mem_t<int>((size_t)m_local + 8, ctx);
        MGX_HIP(hipMemcpyAsync(st.col_pad.data(), st.col_indices, (size_t)m_local * sizeof(int), hipMemcpyDeviceToDevice, ctx.stream()));
        MGX_HIP(hipMemsetAsync(st.col_pad.data() + m_local, 0xFF, 8 * sizeof(int), ctx.stream()));
        st.front_local = mem_t<unsigned>((size_t)n / 32 + 64, ctx);
        MGX_HIP(hipMemsetAsync(st.front_local.data(), 0, st.front_local.size() * sizeof(unsigned), ctx.stream()));
        ctx.synchronize();
        st.vs_div = st.vs_edges ? (unsigned)vdiv : 0u;
      }
    }
  }
  if (units) *units = U;
  // cold-edge lists of the same rows (bfs_fused_cold.hpp; MGX_DIST_COLD=0: none): as build_cold_lists does for a graph's layout --
  // only when the destinations behind the LDS prefix span at most 128 slices of which at most BFS_COLD_MAX_SLICES hold pairs,
  // and the cold entries are at most half of the long rows' entries
  bool want_cold = true;
  if (const char* e = mgx::env("MGX_DIST_COLD")) want_cold = atoi(e) != 0;
  const unsigned hot_n = (unsigned)mgx::BFS_COLD_WORDS * 32u, slice_n = hot_n;
  const long long slices_ll = (unsigned)st.n_global > hot_n ? ((long long)st.n_global - hot_n + slice_n - 1) / slice_n : 0;
  if (want_cold && slices_ll >= 1 && slices_ll <= 128) {
    standard_context_t& ctx = *h->c->ctx;
    mem_t<int> d_long = mgx::fill<int>(0, 1, ctx), d_sorted = mgx::fill<int>(1, 1, ctx);
    hipLaunchKernelGGL(mgx::k_d2_row_facts, dim3(ctx.num_cus * 8), dim3(mgx::BLOCK), 0, ctx.stream(), st.row_offsets, st.col_indices, st.n_local, long_min,
                       d_long.data(), d_sorted.data());
    ctx.synchronize();
    const int long_rows = mgx::from_mem(d_long)[0];
    const bool rows_sorted = mgx::from_mem(d_sorted)[0] == 1;
    const int slices = (int)slices_ll;
    if (rows_sorted && long_rows > 0) {
      std::vector<int> off((size_t)slices + 1, 0);
      int *cowner = nullptr, *cdst = nullptr;
      long long pairs = 0;
      const int rc2 = mgx_cold_build_device(st.row_offsets, st.col_indices, st.n_local, 0, long_rows, long_min, hot_n, slice_n, slices, &cowner, &cdst,
                                            &pairs, off.data(), ctx.stream());
      if (rc2 != 0) throw mgx::mgx_error(MGX_E_HIP, std::string("partitioned BFS, cold-edge lists: ") + hipGetErrorString((hipError_t)rc2));
      // (owned by the state from here on: freed with it whatever happens below; the pass is enabled by its flush buffers)
      st.cold_owner = cowner; st.cold_dst = cdst;
      int used = 0;
      for (int k = 0; k < slices; ++k) if (off[k + 1] > off[k]) ++used;
      if (pairs > 0 && pairs * 2 <= U * 64 && used <= mgx::BFS_COLD_MAX_SLICES) {      // (RMAT-25 / 8: a quarter of the entries, RMAT-26 / 8: a third)
        hipLaunchKernelGGL(mgx::k_d2_owner_global, dim3((unsigned)((pairs + 256 + mgx::BLOCK - 1) / mgx::BLOCK)), dim3(mgx::BLOCK), 0, ctx.stream(), cowner,
                           pairs + 256, st.ranks, st.rank, st.n_local, st.n_global);
        int q = 0;
        for (int k = 0; k < slices; ++k) {
          if (off[k + 1] <= off[k]) continue;
          st.cold_lo[q] = hot_n + (unsigned)k * slice_n;
          st.cold_off[q] = (unsigned)off[k]; st.cold_off[q + 1] = (unsigned)off[k + 1];
          ++q;
        }
        long long nwg = (pairs + 131071) / 131072;
        nwg = std::max<long long>(nwg, mgx::BFS_COLD_WGS);
        // (at most 512 on a rank: every cold workgroup costs a copy of its slice into LDS and an 80 KB bitmap to write and to reduce --
        //  RMAT-26 / 8 with 1 024 of them: push 584 us and reduce 87 us per traversal, with 512: 552 and 57, with 256: 575 and 46)
        nwg = std::min<long long>(nwg, 512);
        if (const char* e = mgx::env("MGX_DIST_COLD_WGS")) if (atoi(e) > 0) nwg = std::min<long long>(atoi(e), mgx::BFS_COLD_WGS_MAX);      // (measurements)
        nwg = std::max<long long>(nwg, used);
        unsigned left = (unsigned)nwg - (unsigned)used, acc = 0;
        st.cold_wgs[0] = 0;
        for (int i = 0; i < used; ++i) {
          const long long cnt = (long long)st.cold_off[i + 1] - (long long)st.cold_off[i];
          unsigned extra = (unsigned)((cnt * (long long)((unsigned)nwg - (unsigned)used)) / pairs);
          if (extra > left) extra = left;
          left -= extra;
          acc += 1u + extra;
          st.cold_wgs[i + 1] = acc;
        }
        for (int i = used + 1; i <= mgx::BFS_COLD_MAX_SLICES; ++i) { st.cold_wgs[i] = acc; st.cold_off[i] = st.cold_off[used]; }
        st.cold_flush = mem_t<u32>((size_t)acc * mgx::BFS_COLD_WORDS, ctx);
        MGX_HIP(hipMemsetAsync(st.cold_flush.data(), 0, (size_t)acc * mgx::BFS_COLD_WORDS * sizeof(unsigned), ctx.stream()));
        // a launch of its own ORs a slice's bitmaps together in front of the sweep (MGX_DIST_COLD_REDUCE=0: k_d2_newbits reads them all)
        if (const char* e = mgx::env("MGX_DIST_COLD_REDUCE")) st.cold_reduce = atoi(e);
        ctx.synchronize();
        st.cold_pairs = pairs; st.cold_slices = used;
        // the pairs once more, four bytes each (owners are global ids by now: `ranks` apart inside a list)
        {
          bool pack_pairs = true;
          if (const char* e = mgx::env("MGX_BFS_COLD_PACK")) pack_pairs = atoi(e) != 0;
          if (pack_pairs) {
            unsigned *pk = nullptr, *cbase = nullptr;
            unsigned long long mask = 0;
            const int rcp = mgx_cold_pack_device(st.cold_owner, st.cold_dst, used, st.cold_off, st.cold_lo, st.ranks, &pk, &cbase, st.cold_cb, &mask, ctx.stream());
            if (rcp != 0) throw mgx::mgx_error(MGX_E_HIP, std::string("partitioned BFS, packed cold-edge lists: ") + hipGetErrorString((hipError_t)rcp));
            st.cold_pk = pk; st.cold_cbase = cbase; st.cold_pk_mask = mask;
          }
        }
        // The unit blocks again, WITHOUT the entries that now live in the pair lists (the unit-block body read them only to skip
        // them: a third of its stream on RMAT-26 / 8) -- and what is left points into the LDS prefix, ids below 2^20: three bytes
        // per entry do (bfs_fused_dense.hpp: ub_col24).  MGX_DIST_HOT_UNITS=0: the full blocks stay.
        bool hot_units = true;
        if (const char* e = mgx::env("MGX_DIST_HOT_UNITS")) hot_units = atoi(e) != 0;
        if (hot_units) {
          int *owner2 = nullptr, *ucol2 = nullptr, *ufirst2 = nullptr;
          unsigned char* ucnt2 = nullptr;
          long long U2 = 0, Up2 = 0;
          const int rc3 = mgx_units_build_device(st.row_offsets, st.col_indices, st.n_local, long_min, 0x7FFFFFFF, 6, hot_n, &owner2, &ucol2, &ucnt2,
                                                 &ufirst2, &U2, &Up2, ctx.stream());
          if (rc3 != 0) throw mgx::mgx_error(MGX_E_HIP, std::string("partitioned BFS, unit blocks of the hot entries: ") + hipGetErrorString((hipError_t)rc3));
          if (ucnt2) (void)hipFree(ucnt2);
          if (ufirst2) (void)hipFree(ufirst2);
          if (U2 > 0) {
            hipLaunchKernelGGL(mgx::k_d2_owner_global, dim3((unsigned)((Up2 + mgx::BLOCK - 1) / mgx::BLOCK)), dim3(mgx::BLOCK), 0, ctx.stream(), owner2, Up2,
                               st.ranks, st.rank, st.n_local, st.n_global);
            ctx.synchronize();
            (void)hipFree(st.ub_owner); (void)hipFree(st.ub_col);
            st.ub_owner = owner2; st.ub_col = ucol2; st.ub_units = U2; st.ub_units_pad = Up2;
            if (units) *units = U2;
            bool pack = true;
            if (const char* e = mgx::env("MGX_BFS_PACK24")) pack = atoi(e) != 0;
            if (pack) {
              const long long quads = ((long long)Up2 << 4) + 1;
              st.ub_col24 = mem_t<unsigned>((size_t)quads * 3 + 4, ctx);
              hipLaunchKernelGGL(k_pack24, dim3(mgx::grid_for(quads, 256, 16384)), dim3(256), 0, ctx.stream(), (const int4*)st.ub_col, quads,
                                 st.ub_col24.data());
              ctx.synchronize();
            }
          }
        }
      } else {
        if (cowner) (void)hipFree(cowner);
        if (cdst) (void)hipFree(cdst);
        st.cold_owner = nullptr; st.cold_dst = nullptr;
      }
    }
  }
  MGX_CATCH
}
int mgx_dbfs2_dense_levels(mgx_dbfs2_t h, int64_t* levels) {
  MGX_TRY
  MGX_REQUIRE(h && levels, "NULL argument");
  *levels = (int64_t)h->st.fs->host_ctrl->dense_slots;     // (as of the last mgx_dbfs2_status / mgx_dbfs2_run)
  MGX_CATCH
}
int mgx_dbfs2_cold_levels(mgx_dbfs2_t h, int64_t* levels, int64_t* pairs) {
  MGX_TRY
  MGX_REQUIRE(h && levels, "NULL argument");
  *levels = (int64_t)h->st.fs->host_ctrl->cold_slots;
  if (pairs) *pairs = (int64_t)h->st.cold_pairs;
  MGX_CATCH
}
int mgx_dbfs2_path_levels(mgx_dbfs2_t h, int64_t* out4) {
  MGX_TRY
  MGX_REQUIRE(h && out4, "NULL argument");
  const mgx::bfs_ctrl_t* hc = h->st.fs->host_ctrl;          // (as of the last mgx_dbfs2_status / mgx_dbfs2_run)
  out4[0] = (int64_t)hc->small_levels;                       // levels whose push appended its discoveries to the id list itself
  out4[1] = (int64_t)hc->vshort_slots;                       // levels whose short rows were walked vertex by vertex
  out4[2] = (int64_t)hc->d2_declared_level >= 0 ? 1 : 0;     // a sweep declared its list overflowed (the last such level is kept, not a count)
  out4[3] = (int64_t)(h->st.cold_pk ? __builtin_popcountll(h->st.cold_pk_mask) : 0);   // slices whose pairs are packed to 4 bytes
  MGX_CATCH
}
int mgx_dbfs2_free(mgx_dbfs2_t h) {
  MGX_TRY
  if (h) { use_device(h->c); delete h; }
  MGX_CATCH
}
int mgx_dbfs2_reset(mgx_dbfs2_t h, int src) {
  MGX_TRY
  MGX_REQUIRE(h && src >= 0 && src < h->st.n_global, "mgx_dbfs2_reset: bad argument");
  use_device(h->c);
  mgx::d2_reset(h->st, src, *h->c->ctx);
  MGX_CATCH
}
int mgx_dbfs2_push(mgx_dbfs2_t h, int level) {
  MGX_TRY
  MGX_REQUIRE(h && level >= 0, "bad argument");
  use_device(h->c);
  mgx::d2_push(h->st, level, *h->c->ctx);
  MGX_CATCH
}
int mgx_dbfs2_merge(mgx_dbfs2_t h, int level, const unsigned* d_gathered) {
  MGX_TRY
  MGX_REQUIRE(h && d_gathered && level >= 0, "bad argument");
  use_device(h->c);
  mgx::d2_merge(h->st, level, d_gathered, h->st.ranks, h->st.nwords, *h->c->ctx);
  MGX_CATCH
}
int mgx_dbfs2_merge_maps(mgx_dbfs2_t h, int level, const unsigned* d_maps, int maps, int64_t stride_words) {
  MGX_TRY
  MGX_REQUIRE(h && d_maps && level >= 0, "bad argument");
  MGX_REQUIRE(maps >= 1 && maps <= 64 && stride_words >= h->st.nwords && stride_words % 4 == 0,
              "mgx_dbfs2_merge_maps: maps must be 1..64 and the stride a multiple of 4 words, at least a bitmap long");
  use_device(h->c);
  mgx::d2_merge(h->st, level, d_maps, maps, (long long)stride_words, *h->c->ctx);
  MGX_CATCH
}
int mgx_dbfs2_or_maps(mgx_dbfs2_t h, const unsigned* d_maps, int maps, int64_t stride_words, int64_t words,
                      unsigned* d_out) {
  MGX_TRY
  MGX_REQUIRE(h && d_maps && d_out, "NULL argument");
  MGX_REQUIRE(maps >= 1 && maps <= 64 && words >= 0 && words % 4 == 0 && stride_words % 4 == 0 && stride_words >= words,
              "mgx_dbfs2_or_maps: maps must be 1..64, words and stride multiples of 4, stride >= words");
  use_device(h->c);
  if (words)
    hipLaunchKernelGGL(mgx::k_d2_or_maps, dim3(mgx::grid_for(words / 4, mgx::BLOCK, 1024)), dim3(mgx::BLOCK), 0,
                       h->c->ctx->stream(), (const uint4*)d_maps, maps, (long long)(stride_words / 4), (long long)(words / 4),
                       (uint4*)d_out);
  MGX_CATCH
}
extern "C" int mgx_shard_plan_device(int scale, int edgefactor, unsigned long long seed, int ranks, int rank, void** handle, int* n_local,
                                     long long* m_local, hipStream_t stream);
extern "C" int mgx_shard_fill_device(void* handle, int* row_offsets, int* col, int* new_of_old, int* old_of_new, int* deg_of_new,
                                     hipStream_t stream);
extern "C" void mgx_shard_free_device(void* handle);
int mgx_dbfs2_shard_plan(mgx_ctx_t c, int scale, int edgefactor, uint64_t seed, int ranks, int rank, void** plan, int* n_local, int64_t* m_local) {
  MGX_TRY
  MGX_REQUIRE(c && plan && n_local && m_local, "NULL argument");
  MGX_REQUIRE(scale >= 1 && scale <= 30 && edgefactor >= 1 && ranks >= 1 && ranks <= 64 && rank >= 0 && rank < ranks,
              "mgx_dbfs2_shard_plan: bad argument");
  use_device(c);
  long long m = 0;
  const int rc = mgx_shard_plan_device(scale, edgefactor, (unsigned long long)seed, ranks, rank, plan, n_local, &m, c->ctx->stream());
  if (rc != 0) throw mgx::mgx_error(MGX_E_HIP, std::string("mgx_dbfs2_shard_plan: ") + hipGetErrorString((hipError_t)rc));
  MGX_REQUIRE(m < (1ll << 31), "mgx_dbfs2_shard_plan: more than 2^31 - 1 entries on one rank (int32 local offsets)");
  *m_local = m;
  MGX_CATCH
}
int mgx_dbfs2_shard_fill(mgx_ctx_t c, void* plan, int* d_row_offsets_local, int* d_col_indices, int* d_new_of_old, int* d_old_of_new,
                         int* d_degree_of_new) {
  MGX_TRY
  MGX_REQUIRE(c && plan && d_row_offsets_local, "NULL argument");
  use_device(c);
  const int rc = mgx_shard_fill_device(plan, d_row_offsets_local, d_col_indices, d_new_of_old, d_old_of_new, d_degree_of_new, c->ctx->stream());
  if (rc != 0) throw mgx::mgx_error(MGX_E_HIP, std::string("mgx_dbfs2_shard_fill: ") + hipGetErrorString((hipError_t)rc));
  MGX_CATCH
}
int mgx_dbfs2_shard_free(void* plan) {
  MGX_TRY
  mgx_shard_free_device(plan);
  MGX_CATCH
}
int mgx_dbfs2_list_words(int n_global, int ranks, int64_t* words) {
  MGX_TRY
  MGX_REQUIRE(words && n_global > 0 && ranks >= 1 && ranks <= 64, "mgx_dbfs2_list_words: bad argument");
  *words = (int64_t)mgx::D2_LIST_HEAD + (int64_t)mgx::d2_state_t::default_list_cap(n_global, ranks);
  MGX_CATCH
}
int mgx_dbfs2_set_list(mgx_dbfs2_t h, unsigned* d_list, int64_t words) {
  MGX_TRY
  MGX_REQUIRE(h, "NULL argument");
  MGX_REQUIRE(!d_list || (words > mgx::D2_LIST_HEAD && words % 4 == 0 && words < (1ll << 31)),
              "mgx_dbfs2_set_list: the list must hold its 4 header words and a multiple of 4 words in all");
  use_device(h->c);
  h->c->ctx->synchronize();
  h->st.set_list(d_list, d_list ? (unsigned)(words - mgx::D2_LIST_HEAD) : 0u);
  MGX_CATCH
}
int mgx_dbfs2_apply_lists(mgx_dbfs2_t h, int level, const unsigned* d_lists, int lists, int64_t stride_words, int64_t* out3) {
  MGX_TRY
  MGX_REQUIRE(h && d_lists && out3 && level >= 0, "bad argument");
  MGX_REQUIRE(h->st.mylist != nullptr, "mgx_dbfs2_apply_lists: no list was set (mgx_dbfs2_set_list)");
  MGX_REQUIRE(lists >= 1 && lists <= 64 && stride_words >= h->st.list_words(), "mgx_dbfs2_apply_lists: 1..64 lists, each at least a list long");
  use_device(h->c);
  long long o[3];
  mgx::d2_apply_lists(h->st, level, d_lists, lists, (long long)stride_words, *h->c->ctx, o);
  for (int i = 0; i < 3; ++i) out3[i] = o[i];
  MGX_CATCH
}
int mgx_dbfs2_status(mgx_dbfs2_t h, int next_level, int64_t* out6) {
  MGX_TRY
  MGX_REQUIRE(h && out6 && next_level >= 0, "bad argument");
  use_device(h->c);
  long long o[6];
  mgx::d2_status(h->st, next_level, *h->c->ctx, o);
  for (int i = 0; i < 6; ++i) out6[i] = o[i];
  MGX_CATCH
}
// ---- RCCL communicator of the library's own (mgx/comm.hpp) and the traversal driven from C++ ----
// (MGX_COMM=loopback, an environment route to the in-process stand-in, is gone since round 6: a loopback world is asked for by id,
//  mgx_comm_loopback_id -- the tests' way)
int mgx_comm_unique_id(unsigned char* out128) {
  MGX_TRY
  MGX_REQUIRE(out128, "NULL argument");
  mgx::rccl_api_t& api = mgx::rccl_api_t::get();
  MGX_REQUIRE(api.ok(), "mgx_comm_unique_id: no RCCL in this process and none could be loaded (librccl.so.1)");
  ncclUniqueId id;
  MGX_RCCL(api.GetUniqueId(&id));
  memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
  MGX_CATCH
}
int mgx_comm_loopback_id(unsigned char* out128) {
  MGX_TRY
  MGX_REQUIRE(out128, "NULL argument");
  const mgx::rccl_api_t& api = mgx::rccl_api_t::loopback();
  MGX_REQUIRE(api.ok(), "mgx_comm_loopback_id: the in-process stand-in for RCCL is a test library of its own (mini_amd/libmgx_loopback.so, built by __graft_entry__.build()): not found next to this library");
  ncclUniqueId id;
  MGX_RCCL(api.GetUniqueId(&id));
  memcpy(out128, id.internal, NCCL_UNIQUE_ID_BYTES);
  MGX_CATCH
}
int mgx_comm_create(mgx_ctx_t c, int ranks, int rank, const unsigned char* id128, mgx_comm_t* out) {
  MGX_TRY
  MGX_REQUIRE(c && id128 && out && ranks >= 1 && rank >= 0 && rank < ranks, "mgx_comm_create: bad argument");
  // (a loopback id -- mgx_comm_loopback_id -- names the in-process stand-in: a test library of its own since round 6)
  const bool loop = mgx::loopback::is_loopback_id(id128);
  const mgx::rccl_api_t& api = loop ? mgx::rccl_api_t::loopback() : mgx::rccl_api_t::get();
  MGX_REQUIRE(api.ok(), loop ? "mgx_comm_create: a loopback id, but mini_amd/libmgx_loopback.so (the test double) is not next to this library"
                             : "mgx_comm_create: no RCCL in this process and none could be loaded (librccl.so.1)");
  use_device(c);
  ncclUniqueId id;
  memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  auto* h = new mgx_comm_s();
  h->c = c;
  h->cm.ranks = ranks; h->cm.rank = rank; h->cm.api = &api;
  ncclResult_t r = api.CommInitRank(&h->cm.comm, ranks, id, rank);
  if (r != ncclSuccess) {
    h->cm.comm = nullptr;
    delete h;
    throw mgx::mgx_error(MGX_E_HIP, std::string("ncclCommInitRank: ") + (api.GetErrorString ? api.GetErrorString(r) : "RCCL error"));
  }
  *out = h;
  MGX_CATCH
}
int mgx_comm_info(mgx_comm_t h, int* is_loopback, int64_t* rounds) {
  MGX_TRY
  MGX_REQUIRE(h, "NULL argument");
  const bool loop = h->cm.api == &mgx::rccl_api_t::loopback();
  if (is_loopback) *is_loopback = loop ? 1 : 0;
  if (rounds) *rounds = (loop && h->cm.api->rounds) ? (int64_t)h->cm.api->rounds(h->cm.comm) : 0;
  MGX_CATCH
}
int mgx_comm_selftest(mgx_comm_t h, const unsigned* d_send, unsigned* d_gathered, unsigned* d_alltoall, int64_t words) {
  MGX_TRY
  MGX_REQUIRE(h && d_send && d_gathered && d_alltoall && words > 0, "mgx_comm_selftest: bad argument");
  use_device(h->c);
  const mgx::rccl_api_t& api = h->cm.table();
  hipStream_t s = h->c->ctx->stream();
  const int R = h->cm.ranks;
  MGX_RCCL(api.AllGather(d_send, d_gathered, (size_t)words, ncclUint32, h->cm.comm, s));
  {
    mgx::rccl_group_t group(api);
    for (int r = 0; r < R; ++r) {
      MGX_RCCL(api.Send(d_send + (size_t)r * words, (size_t)words, ncclUint32, r, h->cm.comm, s));
      MGX_RCCL(api.Recv(d_alltoall + (size_t)r * words, (size_t)words, ncclUint32, r, h->cm.comm, s));
    }
    group.end();
  }
  MGX_HIP(hipStreamSynchronize(s));
  MGX_CATCH
}
int mgx_comm_free(mgx_comm_t h) {
  MGX_TRY
  if (h) { use_device(h->c); delete h; }
  MGX_CATCH
}
const char* mgx_comm_library(void) { return mgx::rccl_api_t::get().where.c_str(); }
int mgx_env_switches(int index, const char** name, const char** what) {
  int n = 0;
  const mgx::env_switch_t* t = mgx::env_switches(&n);
  if (index >= 0 && index < n) { if (name) *name = t[index].name; if (what) *what = t[index].what; }
  return n;
}
int mgx_build_is_lab(void) {
#ifdef MGX_LAB
  return 1;
#else
  return 0;
#endif
}
int mgx_comm_available(void) { return mgx::rccl_api_t::get().ok() ? 1 : 0; }
int mgx_dbfs2_run(mgx_dbfs2_t h, mgx_comm_t comm, int src_global, int exchange, int64_t exchange_words, int64_t* out6) {
  MGX_TRY
  MGX_REQUIRE(h && out6 && src_global >= 0 && src_global < h->st.n_global, "mgx_dbfs2_run: bad argument");
  MGX_REQUIRE(exchange == 0 || exchange == 1, "mgx_dbfs2_run: exchange is 0 (all-gather) or 1 (slices + all-gather)");
  MGX_REQUIRE(comm || h->st.ranks == 1, "mgx_dbfs2_run: a communicator is needed for more than one rank");
  MGX_REQUIRE(!comm || (comm->cm.ranks == h->st.ranks && comm->cm.rank == h->st.rank), "mgx_dbfs2_run: communicator and engine disagree on the partition");
  MGX_REQUIRE(exchange_words >= h->st.nwords && exchange_words % (4 * h->st.ranks) == 0,
              "mgx_dbfs2_run: exchange_words must cover the bitmap and be a multiple of 4 * ranks");
  use_device(h->c);
  mgx::comm_t none;
  long long o[6];
  mgx::d2_run(h->st, comm ? comm->cm : none, h->run_bufs, src_global, exchange, (long long)exchange_words, *h->c->ctx, o);
  for (int i = 0; i < 6; ++i) out6[i] = o[i];
  MGX_CATCH
}
int mgx_dbfs2_spec_stats(mgx_dbfs2_t h, int64_t* out5) {
  MGX_TRY
  MGX_REQUIRE(h && out5, "NULL argument");
  out5[0] = h->run_bufs.spec_runs; out5[1] = h->run_bufs.spec_frozen; out5[2] = h->run_bufs.spec_short;
  out5[3] = h->run_bufs.last_plan_levels; out5[4] = (int64_t)h->run_bufs.last_plan_sparse;
  MGX_CATCH
}
int mgx_dbfs2_forget_plan(mgx_dbfs2_t h) {
  MGX_TRY
  MGX_REQUIRE(h, "NULL argument");
  h->run_bufs.hist_n = 0;
  MGX_CATCH
}
int mgx_dbfs2_run_group(mgx_dbfs2_t* hs, int count, int src_global, int64_t exchange_words, int64_t* out6_each) {
  MGX_TRY
  MGX_REQUIRE(hs && out6_each && count >= 1 && count <= 64, "mgx_dbfs2_run_group: 1..64 engines");
  mgx_dbfs2_t h0 = hs[0];
  MGX_REQUIRE(h0 && count == h0->st.ranks, "mgx_dbfs2_run_group: one engine per rank of the partition");
  MGX_REQUIRE(src_global >= 0 && src_global < h0->st.n_global, "mgx_dbfs2_run_group: bad source");
  MGX_REQUIRE(exchange_words >= h0->st.nwords && exchange_words % 4 == 0, "mgx_dbfs2_run_group: exchange_words must cover the bitmap");
  std::vector<mgx::d2_state_t*> sts((size_t)count);
  std::vector<mgx::d2_run_bufs_t*> bufs((size_t)count);
  for (int r = 0; r < count; ++r) {
    MGX_REQUIRE(hs[r] && hs[r]->c == h0->c && hs[r]->st.ranks == count && hs[r]->st.rank == r && hs[r]->st.n_global == h0->st.n_global &&
                (hs[r]->st.mylist != nullptr) == (h0->st.mylist != nullptr),
                "mgx_dbfs2_run_group: the engines must be ranks 0 .. count - 1 of one partition, on one context, all with or all without id lists");
    sts[(size_t)r] = &hs[r]->st; bufs[(size_t)r] = &hs[r]->run_bufs;
  }
  use_device(h0->c);
  std::vector<long long> o((size_t)count * 6, 0);
  mgx::d2_group_run(sts.data(), bufs.data(), h0->group_bufs, count, src_global, (long long)exchange_words, *h0->c->ctx, o.data());
  for (size_t i = 0; i < o.size(); ++i) out6_each[i] = o[i];
  MGX_CATCH
}
int mgx_dbfs2_words(int n_global, int64_t* words) {
  MGX_TRY
  MGX_REQUIRE(words && n_global > 0, "bad argument");
  *words = (((long long)n_global + 31) / 32 + 3) / 4 * 4;
  MGX_CATCH
}
int mgx_dbfs2_labels(mgx_dbfs2_t h, int* host_labels_local) {
  MGX_TRY
  MGX_REQUIRE(h && host_labels_local, "NULL argument");
  use_device(h->c);
  h->c->ctx->synchronize();
  MGX_HIP(mgx::dtoh(host_labels_local, h->st.labels.data(), (size_t)h->st.n_local));
  MGX_CATCH
}
int mgx_dbfs2_visited(mgx_dbfs2_t h, unsigned* host_words) {
  MGX_TRY
  MGX_REQUIRE(h && host_words, "NULL argument");
  use_device(h->c);
  h->c->ctx->synchronize();
  MGX_HIP(mgx::dtoh(host_words, h->st.fs->visited.data(), (size_t)h->st.nwords));
  MGX_CATCH
}

// ---- SSSP ----------------------------------------------------------------------------------
static void check_weights(mgx_graph_t g) {
  if (g->weights_checked) return;
  standard_context_t& ctx = *g->c->ctx;
  mem_t<int> flag = mgx::fill<int>(0, 1, ctx);
  int* pf = flag.data();
  const float* w = g->g->d_col_values.data();
  mgx::transform([=] __device__(int i) { if (!(w[i] >= 0.0f)) *pf = 1; }, g->g->num_edges, ctx);
  ctx.synchronize();
  g->weights_ok = (mgx::from_mem(flag)[0] == 0);
  g->weights_checked = true;
}

int mgx_sssp_create(mgx_graph_t g, int src, mgx_sssp_t* out) {
  MGX_TRY
  MGX_REQUIRE(g && out, "NULL argument");
  MGX_REQUIRE(src >= 0 && src < g->g->num_nodes, "mgx_sssp_create: src out of range");
  use_device(g->c);
  check_weights(g);
  if (!g->weights_ok) throw mgx::mgx_error(MGX_E_NEGATIVE_WEIGHT, "mgx_sssp_create: negative or NaN edge weight");
  auto* h = new mgx_sssp_s();
  h->g = g;
  h->p = std::make_shared<sssp::sssp_problem_t>(g->g, (size_t)src, *g->c->ctx);
  *out = h;
  MGX_CATCH
}
int mgx_sssp_reset(mgx_sssp_t p, int src) {
  MGX_TRY
  MGX_REQUIRE(p, "NULL argument");
  MGX_REQUIRE(src >= 0 && src < p->g->g->num_nodes, "mgx_sssp_reset: src out of range");
  use_device(p->g->c);
  p->p->reset((size_t)src, *p->g->c->ctx);
  p->preds_stale = false;          // (-1 everywhere: what the operator path starts from)
  MGX_CATCH
}
int mgx_sssp_free(mgx_sssp_t p) {
  MGX_TRY
  if (p) { use_device(p->g->c); delete p; }
  MGX_CATCH
}
int mgx_sssp_distances(mgx_sssp_t p, float* host) {
  MGX_TRY
  MGX_REQUIRE(p && host, "NULL argument");
  use_device(p->g->c);
  p->g->c->ctx->synchronize();
  MGX_HIP(mgx::dtoh(host, p->p->d_labels.data(), (size_t)p->g->g->num_nodes));
  MGX_CATCH
}
// the fused loop keeps no predecessors (sssp_fused.hpp); they are built from its distances the first time they are asked for
static void sssp_preds_if_stale(mgx_sssp_t p) {
  if (!p->preds_stale) return;
  graph_device_t& G = *p->g->g;
  mgx::sssp_build_preds(p->pred_state, G.d_row_offsets.data(), G.d_col_indices.data(), G.d_col_values.data(), p->p->d_labels.data(),
                        p->p->d_preds.data(), G.num_nodes, (long long)G.num_edges, p->preds_src, 3.402823466e+38f, *p->g->c->ctx);
  p->preds_stale = false;
}
int mgx_sssp_build_preds(mgx_sssp_t p, int64_t* stats2) {
  MGX_TRY
  MGX_REQUIRE(p, "NULL argument");
  use_device(p->g->c);
  sssp_preds_if_stale(p);
  if (stats2) { stats2[0] = p->pred_state.last_ties; stats2[1] = p->pred_state.last_rounds; }
  MGX_CATCH
}
int mgx_sssp_preds(mgx_sssp_t p, int* host) {
  MGX_TRY
  MGX_REQUIRE(p && host, "NULL argument");
  use_device(p->g->c);
  sssp_preds_if_stale(p);
  p->g->c->ctx->synchronize();
  MGX_HIP(mgx::dtoh(host, p->p->d_preds.data(), (size_t)p->g->g->num_nodes));
  MGX_CATCH
}
int mgx_sssp_distances_device(mgx_sssp_t p, float** d) {
  MGX_TRY
  MGX_REQUIRE(p && d, "NULL argument");
  *d = p->p->d_labels.data();
  MGX_CATCH
}
int mgx_sssp_advance(mgx_sssp_t p, mgx_frontier_t in, mgx_frontier_t out, int iteration, int64_t* front) {
  MGX_TRY
  MGX_REQUIRE(p && in && out, "NULL argument");
  use_device(p->g->c);
  const int r = oprtr::advance::advance_forward_kernel<sssp::sssp_problem_t, sssp::sssp_functor_t, false, true>(
      p->p, in->f, out->f, iteration, *p->g->c->ctx);
  if (front) *front = r;
  MGX_CATCH
}
int mgx_sssp_filter(mgx_sssp_t p, mgx_frontier_t in, mgx_frontier_t out, int iteration, int64_t* kept) {
  MGX_TRY
  MGX_REQUIRE(p && in && out, "NULL argument");
  use_device(p->g->c);
  const int r = oprtr::filter::filter_kernel<sssp::sssp_problem_t, sssp::sssp_functor_t>(p->p, in->f, out->f,
                                                                                       iteration, *p->g->c->ctx);
  if (kept) *kept = r;
  MGX_CATCH
}
int mgx_sssp_enact(mgx_sssp_t p, float queue_sizing, int64_t* stats) {
  MGX_TRY
  MGX_REQUIRE(p && queue_sizing > 0, "bad argument");
  use_device(p->g->c);
  standard_context_t& ctx = *p->g->c->ctx;
  if (!p->e || p->e_sizing != queue_sizing) {
    p->e.reset();
    p->e.reset(new sssp::sssp_enactor_t(ctx, p->g->g->num_nodes, p->g->g->num_edges, queue_sizing));
    p->e_sizing = queue_sizing;
  }
  p->e->enact(p->p, ctx);
  ctx.synchronize();
  if (stats) {
    stats[0] = p->e->iterations;
    stats[1] = p->e->relaxations;
    stats[2] = p->e->frontier_total;
  }
  MGX_CATCH
}
// Weights of the unit blocks' entries (mgx/sssp_fused.hpp: sssp_dense_long), once per graph at its first fused SSSP run:
// 4 bytes per padded long-row entry; skipped (the queue walk serves every iteration) when the memory is not there.
static void ensure_unit_weights(mgx_graph_s* g) {
  graph_device_t& G = *g->g;
  if (G.ub_w_tried) return;
  G.ub_w_tried = true;
  // (the sweep reads the long rows from the unit blocks and walks the short ones by degree class: any threshold the two were cut by together)
  if (!G.has_layout || !G.has_layout_weights || G.ub_units <= 0 || !G.d_ub_first.size() || G.vs_long_min != G.ub_min_degree || G.vs_long_min < 17 || G.vs_long_min > 64) return;
  if (const char* e = mgx::env("MGX_SSSP_DENSE")) if (atoi(e) == 0) return;
  standard_context_t& ctx = *g->c->ctx;
  float* w = nullptr;
  const size_t entries = ((size_t)G.ub_units_pad << 6) + 4;
  if (hipMalloc((void**)&w, entries * sizeof(float)) != hipSuccess) { (void)hipGetLastError(); return; }
  G.d_ub_w = mem_t<float>::adopt(w, entries);
  MGX_HIP(hipMemsetAsync(w, 0, entries * sizeof(float), ctx.stream()));
  hipLaunchKernelGGL(mgx::k_sssp_unit_weights, dim3(4096), dim3(mgx::BLOCK), 0, ctx.stream(), G.d_layout_row_offsets.data(),
                     G.d_layout_col_values.data(), G.d_ub_first.data(), G.num_nodes, w);
  MGX_CHECK_LAUNCH("unit-block weights");
  // ... and as halves, for the sweep's packed stream: only with the 24-bit entries, only when every weight survives the round trip
  G.d_ub_w16 = mem_t<unsigned short>();
  if (G.d_ub_col24.size()) {
    mem_t<unsigned short> w16(entries + 8, ctx);
    mem_t<int> exact = mgx::fill<int>(1, 1, ctx);
    hipLaunchKernelGGL(mgx::k_sssp_unit_weights16, dim3(4096), dim3(mgx::BLOCK), 0, ctx.stream(), (const float*)w, (long long)entries, w16.data(), exact.data());
    int ok = 0;
    MGX_HIP(mgx::dtoh(&ok, exact.data(), 1));
    if (ok) G.d_ub_w16 = std::move(w16);
  }
}
int mgx_sssp_run(mgx_sssp_t p, int src, int64_t* stats) { return mgx_sssp_run_delta(p, src, -1.0f, stats); }
int mgx_sssp_run_delta(mgx_sssp_t p, int src, float delta, int64_t* stats) {
  // device-resident loop (include/mgx/sssp_fused.hpp): distances identical to mgx_sssp_enact's; predecessors are
  // left at -1 (the reference's are racy, the operator path keeps them)
  int rc = mgx_sssp_reset(p, src);
  if (rc != MGX_OK) return rc;
  MGX_TRY
  use_device(p->g->c);
  standard_context_t& ctx = *p->g->c->ctx;
  graph_device_t& G = *p->g->g;
  MGX_REQUIRE(G.d_col_values.size() >= (size_t)G.num_edges, "mgx_sssp_run: the graph has no weights");
  check_weights(p->g);
  MGX_REQUIRE(p->g->weights_ok, "mgx_sssp_run: negative or NaN weight");
  if (!p->fused) p->fused.reset(new mgx::sssp_fused_state_t(G.num_nodes, ctx));
  MGX_REQUIRE(!(delta != delta) && delta < 3e38f, "mgx_sssp_run_delta: delta must be a finite number (< 0: the default)");
  p->fused->delta = delta < 0.f ? 0.f : delta;
  mgx::sssp_layout_t layout;
  if (G.has_layout && G.has_layout_weights) {
    layout.row_offsets = G.d_layout_row_offsets.data();
    layout.col_indices = G.d_layout_col_indices.data();
    layout.weights = G.d_layout_col_values.data();
    layout.new_of_old = G.d_new_of_old.data();
    layout.old_of_new = G.d_old_of_new.data();
    ensure_unit_weights(p->g);
    if (G.d_ub_w.size()) {
      layout.ub_col = G.d_ub_col.size() ? G.d_ub_col.data() : nullptr; layout.ub_w = G.d_ub_w.data(); layout.ub_cnt = G.d_ub_cnt.data(); layout.ub_owner = G.d_ub_owner.data();
      if (G.d_ub_col24.size()) layout.ub_col24 = G.d_ub_col24.data();
      if (G.d_ub_col24.size() && G.d_ub_w16.size()) layout.ub_w16 = G.d_ub_w16.data();
      layout.ub_units_pad = (unsigned)G.ub_units_pad;
      for (int i = 0; i < 4; ++i) layout.vs_v[i] = G.vs_v[i];
      layout.m_edges = (long long)G.num_edges;
    }
  }
  mgx::sssp_fused_run(*p->fused, G.d_row_offsets.data(), G.d_col_indices.data(), G.d_col_values.data(),
                      p->p->d_labels.data(), src, ctx, layout.row_offsets ? &layout : nullptr);
  p->preds_stale = true;           // (mgx_sssp_preds / mgx_sssp_build_preds: from these distances, mgx/sssp_preds.hpp)
  p->preds_src = src;
  if (stats) {
    stats[0] = p->fused->host_ctrl->levels;
    stats[1] = (int64_t)p->fused->host_ctrl->sum_edges;
    stats[2] = (int64_t)p->fused->host_ctrl->sum_frontier;
  }
  MGX_CATCH
}
int mgx_sssp_iteration_trace(mgx_sssp_t p, int cap, int64_t* frontier, int64_t* edges, float* ms, int* iterations) {
  MGX_TRY
  MGX_REQUIRE(p && iterations, "NULL argument");
  MGX_REQUIRE(p->fused != nullptr, "mgx_sssp_iteration_trace: no mgx_sssp_run yet");
  const mgx::bfs_ctrl_t* hc = p->fused->host_ctrl;
  const int it = hc->levels < 63 ? hc->levels : 63;
  *iterations = hc->levels;
  for (int i = 0; i < it && i < cap; ++i) {
    if (frontier) frontier[i] = (int64_t)(hc->trace[i] >> mgx::BFS_VSHIFT);
    if (edges) edges[i] = (int64_t)(hc->trace[i] & mgx::BFS_EMASK);
    if (ms) ms[i] = (hc->stamp[i + 1] >= hc->stamp[i] && hc->stamp[i + 1] != 0) ? (float)((double)(hc->stamp[i + 1] - hc->stamp[i]) / 1e5) : 0.f;
  }
  MGX_CATCH
}
int mgx_sssp_set_kernel_timing(mgx_sssp_t p, int on) {
  MGX_TRY
  MGX_REQUIRE(p, "NULL argument");
  use_device(p->g->c);
  if (!p->fused) p->fused.reset(new mgx::sssp_fused_state_t(p->g->g->num_nodes, *p->g->c->ctx));
  p->fused->time_kernels = on != 0;
  MGX_CATCH
}
int mgx_sssp_kernel_times(mgx_sssp_t p, int64_t* out2) {
  MGX_TRY
  MGX_REQUIRE(p && out2, "NULL argument");
  MGX_REQUIRE(p->fused != nullptr, "mgx_sssp_kernel_times: no mgx_sssp_run yet");
  out2[0] = p->fused->relax_launches;
  out2[1] = (int64_t)(p->fused->relax_ms * 1e6);
  MGX_CATCH
}

// ---- PR ------------------------------------------------------------------------------------
int mgx_pr_create(mgx_graph_t g, int max_iter, mgx_pr_t* out) {
  MGX_TRY
  MGX_REQUIRE(g && out && max_iter >= 0, "bad argument");
  use_device(g->c);
  auto* h = new mgx_pr_s();
  h->g = g;
  h->p = std::make_shared<pr::pr_problem_t>(g->g, max_iter, *g->c->ctx);
  // the enactor with its frontiers (2 x num_edges ints, enactor.hxx:22-28) and the layout's sliced long rows belong to the set-up, as
  // in the reference's driver (tests/pr/test_pr.cu:32 constructs pr_enactor_t before the timer around enact starts, :34-37)
  h->e.reset(new pr::pr_enactor_t(*g->c->ctx, g->g->num_nodes, g->g->num_edges));
  ensure_nr_slices(g);
  *out = h;
  MGX_CATCH
}
int mgx_pr_free(mgx_pr_t p) {
  MGX_TRY
  if (p) { use_device(p->g->c); delete p; }
  MGX_CATCH
}
int mgx_pr_enact(mgx_pr_t p, int64_t* lens, int* iterations) {
  MGX_TRY
  MGX_REQUIRE(p, "NULL argument");
  use_device(p->g->c);
  standard_context_t& ctx = *p->g->c->ctx;
  if (!p->e) p->e.reset(new pr::pr_enactor_t(ctx, p->g->g->num_nodes, p->g->g->num_edges));
  ensure_nr_slices(p->g);
  p->e->enact(p->p, ctx);
  ctx.synchronize();
  if (iterations) *iterations = (int)p->e->frontier_lengths.size();
  if (lens)
    for (size_t i = 0; i < p->e->frontier_lengths.size(); ++i) lens[i] = p->e->frontier_lengths[i];
  MGX_CATCH
}
int mgx_pr_ranks(mgx_pr_t p, float* host) {
  MGX_TRY
  MGX_REQUIRE(p && host, "NULL argument");
  use_device(p->g->c);
  p->g->c->ctx->synchronize();
  MGX_HIP(mgx::dtoh(host, p->p->d_current_ranks.data(), (size_t)p->g->g->num_nodes));
  MGX_CATCH
}


// ---- k-core ----------------------------------------------------------------------------------
int mgx_kcore_create(mgx_graph_t g, mgx_kcore_t* out) {
  MGX_TRY
  MGX_REQUIRE(g && out, "bad argument");
  use_device(g->c);
  auto* h = new mgx_kcore_s();
  h->g = g;
  h->p = std::make_shared<kcore::kcore_problem_t>(g->g, *g->c->ctx);
  *out = h;
  MGX_CATCH
}
int mgx_kcore_reset(mgx_kcore_t p) {
  MGX_TRY
  MGX_REQUIRE(p, "NULL argument");
  use_device(p->g->c);
  p->p->reset(*p->g->c->ctx);
  MGX_CATCH
}
int mgx_kcore_free(mgx_kcore_t p) {
  MGX_TRY
  if (p) { use_device(p->g->c); delete p; }
  MGX_CATCH
}
int mgx_kcore_enact(mgx_kcore_t p, int* largest_k_core, int64_t* stats) {
  MGX_TRY
  MGX_REQUIRE(p && largest_k_core, "NULL argument");
  use_device(p->g->c);
  standard_context_t& ctx = *p->g->c->ctx;
  if (!p->e) p->e.reset(new kcore::kcore_enactor_t(ctx, p->g->g->num_nodes, p->g->g->num_edges));
  p->e->enact(p->p, ctx);
  ctx.synchronize();
  *largest_k_core = p->p->largest_k_core;
  if (stats) {
    stats[0] = p->e->rounds;
    stats[1] = p->e->passes;
    stats[2] = p->e->expanded;
    stats[3] = p->e->removed;
  }
  MGX_CATCH
}
int mgx_kcore_num_cores(mgx_kcore_t p, int* host) {
  MGX_TRY
  MGX_REQUIRE(p && host, "NULL argument");
  use_device(p->g->c);
  p->g->c->ctx->synchronize();
  MGX_HIP(mgx::dtoh(host, p->p->d_num_cores.data(), (size_t)p->g->g->num_nodes));
  MGX_CATCH
}
int mgx_kcore_degrees(mgx_kcore_t p, int* host) {
  MGX_TRY
  MGX_REQUIRE(p && host, "NULL argument");
  use_device(p->g->c);
  p->g->c->ctx->synchronize();
  MGX_HIP(mgx::dtoh(host, p->p->d_degrees.data(), (size_t)p->g->g->num_nodes));
  MGX_CATCH
}

}  // extern "C"

// ---- R-MAT generator (spec: oracle/oracle.c orc_rmat_edges) -----------------------
namespace {
__global__ void k_rmat_edges(int scale, long long first_edge, long long count, unsigned long long seed, int do_scramble,
                             int* __restrict__ src, int* __restrict__ dst, float* __restrict__ weight) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < count; i += stride) {
    const unsigned long long e = (unsigned long long)(first_edge + i);
    unsigned s, d;
    mgx::rmat_pair(scale, seed, e, do_scramble, s, d);
    src[i] = (int)s;
    dst[i] = (int)d;
    if (weight) weight[i] = mgx::rmat_weight(seed, e);
  }
}
}  // namespace

extern "C" int mgx_rmat_edges(mgx_ctx_t c, int scale, int64_t first_edge, int64_t count, uint64_t seed, int scramble_ids,
                              int* d_src, int* d_dst, float* d_weight) {
  MGX_TRY
  MGX_REQUIRE(c && d_src && d_dst && scale >= 1 && scale <= 31 && count >= 0 && first_edge >= 0,
              "mgx_rmat_edges: bad argument");
  use_device(c);
  if (count > 0)
    hipLaunchKernelGGL(k_rmat_edges, dim3(mgx::grid_for(count, 256, 8192)), dim3(256), 0, c->ctx->stream(), scale,
                       (long long)first_edge, (long long)count, (unsigned long long)seed, scramble_ids, d_src, d_dst,
                       d_weight);
  MGX_CATCH
}
