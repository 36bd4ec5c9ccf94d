// gunrock/bfs/bfs_enactor.hxx -- BFS superstep loops.
//   enact_pushpull : the reference's loop, operator per superstep
//                    (gunrock/src/bfs/bfs_enactor.hxx:41-117): push until
//                    num_unvisited < frontier_length * threshold, then the pull phase.
//   enact_fused    : same labels from the device-resident fused path (mgx/bfs_fused.hpp):
//                    one kernel per level, no host round trip per level.
#pragma once
#include <memory>
#include <string>
#include <utility>
#include <vector>
#include "../advance.hxx"
#include "../enactor.hxx"
#include "../filter.hxx"
#include "../frontier.hxx"
#include "../graph.hxx"
#include "../../mgx/bfs_fused_run.hpp"
#include "bfs_functor.hxx"
#include "bfs_problem.hxx"

namespace gunrock {
namespace bfs {

struct bfs_enactor_t : enactor_t {
  // counters of the last enact_* call
  int pushed_iterations = 0;
  int total_iterations = 0;
  long long pushed_edges = 0;   // sum of advance fronts
  long long pulled_edges = 0;   // in-edges inspected by the pull phase
  bool verbose = false;

  bfs_enactor_t(standard_context_t& context, int num_nodes, int num_edges)
      : enactor_t(context, num_nodes, num_edges) {}

  bfs_enactor_t(const bfs_enactor_t& rhs) = delete;
  bfs_enactor_t& operator=(const bfs_enactor_t& rhs) = delete;

  void init_frontier(std::shared_ptr<bfs_problem_t> bfs_problem) {
    std::vector<int> node_idx(1, bfs_problem->src);
    (void)buffers[0]->load(node_idx);
  }

  void enact_pushpull(std::shared_ptr<bfs_problem_t> bfs_problem, float threshold, standard_context_t& context) {
    using namespace gunrock::oprtr::advance;
    using namespace gunrock::oprtr::filter;
    init_frontier(bfs_problem);
    // the index frontiers are consumed by a pull phase; restore them for re-use of the enactor
    unvisited[0] = indices;
    unvisited[1] = filtered_indices;
    const int num_nodes = bfs_problem->gslice->num_nodes;
    indices->resize(num_nodes);
    filtered_indices->resize(num_nodes);

    int frontier_length = 1;
    int selector = 0;
    int num_unvisited = num_nodes - 1;
    int iteration;
    pushed_edges = pulled_edges = 0;

    for (iteration = 0;; ++iteration) {
      frontier_length = advance_forward_kernel<bfs_problem_t, bfs_functor_t, false, true>(
          bfs_problem, buffers[selector], buffers[selector ^ 1], iteration, context);
      pushed_edges += frontier_length;
      selector ^= 1;
      if (!frontier_length) break;
      frontier_length = filter_kernel<bfs_problem_t, bfs_functor_t>(bfs_problem, buffers[selector],
                                                                   buffers[selector ^ 1], iteration, context);
      num_unvisited -= frontier_length;
      if ((float)num_unvisited < (float)frontier_length * threshold) break;
      if (!frontier_length) break;
      selector ^= 1;
    }
    pushed_iterations = total_iterations = iteration;
    if (verbose) std::cout << "pushed iterations: " << iteration << std::endl;

    if (frontier_length) {
      // switch to pull: unvisited list + int-per-vertex bitmap of the current frontier
      ++iteration;
      frontier_length = gen_unvisited_kernel<bfs_problem_t, bfs_functor_t>(bfs_problem, unvisited[selector ^ 1],
                                                                          unvisited[selector], 0, context);
      int new_frontier_length;
      mem_t<int> bitmap_array = mgx::fill<int>(0, num_nodes, context);
      (void)buffers[selector]->load(bitmap_array);
      sparse_to_dense_kernel<bfs_problem_t, bfs_functor_t>(bfs_problem, buffers[selector ^ 1], buffers[selector],
                                                           iteration, context);
      for (;; ++iteration) {
        (void)buffers[selector ^ 1]->load(bitmap_array);
        pulled_edges += advance_backward_kernel<bfs_problem_t, bfs_functor_t>(
            bfs_problem, unvisited[selector], buffers[selector], buffers[selector ^ 1], iteration, context);
        new_frontier_length = filter_kernel<bfs_problem_t, bfs_functor_t>(bfs_problem, unvisited[selector],
                                                                         unvisited[selector ^ 1], iteration, context);
        if (!new_frontier_length || new_frontier_length == frontier_length) break;
        frontier_length = new_frontier_length;
        selector ^= 1;
      }
      total_iterations = iteration;
      if (verbose) std::cout << "total iterations: " << iteration << std::endl;
      // the pull phase leaves -1 holes in the index frontiers: refill the iota for the next run
      mem_t<int> iota = mgx::fill_function<int>([] __device__(int i) { return i; }, num_nodes, context);
      (void)indices->load(iota);
      (void)filtered_indices->load(iota);
    }
  }

  // BFS in the reference's IDEMPOTENT mode (advance.hxx:60 with idempotence = true, filter.hxx:95-119; no upstream
  // enactor instantiates it): advance emits EVERY neighbour of the frontier -- no label test, no atomicCAS per edge --
  // and uniquify keeps one copy of each vertex not seen before (intra-wave cull, then the exact visited-bitmask cull)
  // and labels it (bfs_idempotent_functor_t::cond_uniq).  visited_mask: (num_nodes + 31) / 32 words, cleared here.
  // Same labels as enact_pushpull's push phase; what it trades is the CAS per edge for an int per edge written and read.
  long long idempotent_edges = 0;     // sum of advance fronts of the last enact_idempotent
  void enact_idempotent(std::shared_ptr<bfs_problem_t> bfs_problem, mem_t<unsigned>& visited_mask,
                        standard_context_t& context) {
    using namespace gunrock::oprtr::advance;
    using namespace gunrock::oprtr::filter;
    init_frontier(bfs_problem);
    const int src = bfs_problem->src;
    unsigned* const mask = visited_mask.data();
    MGX_HIP(hipMemsetAsync(mask, 0, visited_mask.size() * sizeof(unsigned), context.stream()));
    mgx::transform([=] __device__(int) { mask[src >> 5] = 1u << (src & 31); }, 1, context);
    idempotent_edges = 0;
    int selector = 0, iteration;
    for (iteration = 0;; ++iteration) {
      const int front = advance_forward_kernel<bfs_problem_t, bfs_idempotent_functor_t, true, true>(
          bfs_problem, buffers[selector], buffers[selector ^ 1], iteration, context);
      idempotent_edges += front;
      if (!front) break;
      uniquify_kernel<bfs_problem_t, bfs_idempotent_functor_t>(bfs_problem, (unsigned char*)mask, buffers[selector ^ 1],
                                                               buffers[selector], iteration, context);
      if (!buffers[selector]->size()) { ++iteration; break; }
    }
    pushed_iterations = total_iterations = iteration;
  }
};

// Device-resident fused traversal (mgx/bfs_fused*.hpp).  Needs only O(n) state: no edge-capacity
// ping-pong buffers.  Counters of the last run are in `last`.
struct bfs_kernel_stats_t {
  long long launches = 0, ns = 0, edges = 0, vertices = 0;   // launches incl. the ones that find nothing to do
};
struct bfs_run_stats_t {
  int levels = 0, push_levels = 0;
  long long reached = 0, m_t = 0, push_edges = 0, pull_edges = 0, frontier_vertices = 0, claims = 0;
  long long kernel_launches = 0, kernel_ns = 0;          // all level kernels of the run (batch events)
  // the two push kernels, timed per launch: row-wise streaming of the long-row queue, per-edge search over
  // the short-row queue; and which of them took longer (1 = stream, 0 = wave)
  bfs_kernel_stats_t stream, wave;
  int dominant = 0;
  std::vector<std::pair<long long, long long>> trace;   // (frontier vertices, frontier edges) per level
  std::vector<float> batch_ms;
  std::vector<float> level_ms;       // device-side stamps (100 MHz) taken when each level is opened
  int small_levels = 0;              // levels run inside a push launch's block 0 (chains of small levels)
  int slots = 0;                     // launch slots the run used
  int dense_slots = 0;               // slots whose long rows were read from the unit blocks
  int vshort_slots = 0;              // slots whose short rows were walked vertex by vertex
  int lazy_slots = 0;                // slots that ran without queues (bfs_build_is_lazy)
  int cold_slots = 0;                // slots that ran the cold-edge pass (bfs_fused_cold.hpp)
  int mini_slots = 0;                // levels expanded by M launches (bfs_fused_mini.hpp)
  long long claims_level[64] = {0};
};

struct bfs_fused_enactor_t {
  std::unique_ptr<mgx::bfs_fused_state_t> fused;
  bfs_run_stats_t last;

  bfs_fused_enactor_t(standard_context_t& context, int num_nodes) {
    fused.reset(new mgx::bfs_fused_state_t(num_nodes, context));
  }
  bfs_fused_enactor_t(const bfs_fused_enactor_t&) = delete;
  bfs_fused_enactor_t& operator=(const bfs_fused_enactor_t&) = delete;

  // the graph's hub-first layout (if it has one) as the fused loop takes it
  static mgx::bfs_layout_t layout_of(graph_device_t& g) {
    mgx::bfs_layout_t layout;
    if (g.has_layout) {
      layout.row_offsets = g.d_layout_row_offsets.data();
      layout.col_indices = g.d_layout_col_indices.data();
      layout.new_of_old = g.d_new_of_old.data();
      layout.old_of_new = g.d_old_of_new.data();
      if (g.ub_units > 0) {
        layout.ub_col = g.d_ub_col.size() ? g.d_ub_col.data() : nullptr;      // (round 6: gone when the 24-bit copy below exists)
        layout.ub_col24 = g.d_ub_col24.size() ? g.d_ub_col24.data() : nullptr;
        layout.ub_owner = g.d_ub_owner.data();
        layout.ub_units = g.ub_units;
        layout.ub_units_pad = g.ub_units_pad;
        layout.ub_min_degree = g.ub_min_degree;
      }
      for (int i = 0; i < 4; ++i) layout.vs_v[i] = g.vs_v[i];
      layout.vs_v9 = g.vs_v9;
      layout.vs_edges = g.vs_edges; layout.vs_dummy = g.vs_dummy; layout.vs_long_min = g.vs_long_min;
      if (g.cold_slices > 0) {
        // (round 6: the 8-byte pairs are gone when every slice carries the packed words below -- cold_pairs8 says which)
        layout.cold_pairs8 = g.d_cold_owner.size() != 0 && g.d_cold_dst.size() != 0;
        layout.cold_owner = layout.cold_pairs8 ? g.d_cold_owner.data() : nullptr;
        layout.cold_dst = layout.cold_pairs8 ? g.d_cold_dst.data() : nullptr;
        layout.cold_slices = g.cold_slices;
        if (g.d_cold_pk.size() && g.d_cold_cbase.size()) {
          layout.cold_pk = g.d_cold_pk.data(); layout.cold_cbase = g.d_cold_cbase.data(); layout.cold_pk_mask = g.cold_pk_mask;
          for (int i = 0; i <= mgx::BFS_COLD_MAX_SLICES; ++i) layout.cold_cb[i] = g.cold_cb[i];
        }
        static_assert(mgx::BFS_COLD_MAX_SLICES == 64, "graph_device_t::cold_* hold this many slices");
        for (int i = 0; i < mgx::BFS_COLD_MAX_SLICES; ++i) layout.cold_lo[i] = g.cold_lo[i];
        for (int i = 0; i <= mgx::BFS_COLD_MAX_SLICES; ++i) { layout.cold_off[i] = g.cold_off[i]; layout.colds_off[i] = g.colds_off[i]; layout.cold_wgs[i] = g.cold_wgs[i]; }
        layout.colds_owner = g.colds_pairs > 0 ? g.d_colds_owner.data() : nullptr;
        layout.colds_dst = g.colds_pairs > 0 ? g.d_colds_dst.data() : nullptr;
        layout.cold_hot_n = g.cold_hot_n;
        layout.cold_long_min = g.cold_long_min;
        if (g.ubh_units > 0 && g.d_ubh_col24.size() && g.d_ubh_owner.size()) {
          layout.ubh_col24 = g.d_ubh_col24.data(); layout.ubh_owner = g.d_ubh_owner.data();
          layout.ubh_units = g.ubh_units; layout.ubh_units_pad = g.ubh_units_pad;
        }
      }
      layout.cold_majority = g.cold_majority;
      layout.cold_all = g.cold_all;
      layout.cold_pairs_total = g.cold_all ? (unsigned long long)g.cold_pairs : 0ull;
    }
    return layout;
  }
  // the shapes of the sources of one call (mgx/src_shapes.hpp), resolved into `table` and hung into the layout: entry i is source i's
  std::vector<unsigned> shape_table;
  void resolve_shapes(graph_device_t& g, mgx::bfs_layout_t& layout, const int* srcs, int count, int mode, standard_context_t& context) {
    if (!g.has_layout || !g.src_shapes_enabled || !mgx::bfs_wants_src_shapes(*fused, mode) || g.num_edges <= 0) return;
    g.src_shape_cache.resolve(g.d_row_offsets.data(), g.d_col_indices.data(), g.num_nodes, fused->long_min, srcs, count, shape_table, context);
    layout.src_shapes = shape_table.data();
    layout.src_shapes_long_min = fused->long_min;
  }

  // counters of a traversal from the (head of the) control block it left on the host
  void fill_stats(bfs_run_stats_t& last, const mgx::bfs_ctrl_t* hc, bool direction_optimizing, bool with_timing) const {
    last.levels = hc->levels; last.push_levels = hc->push_levels;
    last.reached = (long long)(hc->reached + hc->reached_mini); last.m_t = (long long)hc->sum_edges;
    last.pull_edges = (long long)hc->pull_edges; last.frontier_vertices = (long long)hc->sum_frontier;
    last.claims = (long long)hc->claims;
    for (int i = 0; i < hc->levels && i < (with_timing ? mgx::BFS_MAX_TRACE : 64); ++i)
      last.trace.emplace_back((long long)(hc->trace[i] >> mgx::BFS_VSHIFT), (long long)(hc->trace[i] & mgx::BFS_EMASK));
    // (a level's duration needs the stamp of the NEXT level's opening: the direct scheme ends a traversal from the
    //  read-back cursors without opening the level behind the last one -- no time for that last level then)
    for (int i = 0; i < hc->levels && i < 63 && hc->stamp[i + 1] >= hc->stamp[i] && hc->stamp[i + 1] != 0; ++i)
      last.level_ms.push_back((float)((double)(hc->stamp[i + 1] - hc->stamp[i]) / 1e5));
    last.small_levels = hc->small_levels;
    last.slots = fused->slots_used;
    last.dense_slots = hc->dense_slots;
    last.vshort_slots = hc->vshort_slots;
    last.lazy_slots = hc->lazy_slots;
    last.cold_slots = hc->cold_slots;
    last.mini_slots = hc->mini_slots;
    for (int i = 0; i < last.push_levels && i < (int)last.trace.size(); ++i) last.push_edges += last.trace[i].second;
    if (hc->levels > (int)last.trace.size() && !direction_optimizing) last.push_edges = last.m_t;   // (a deep traversal whose trace tail stayed on the device)
    long long push_vertices = 0;
    for (int i = 0; i < last.push_levels && i < (int)last.trace.size(); ++i) push_vertices += last.trace[i].first;
    if (with_timing) {
      last.kernel_launches = fused->level_kernel_launches; last.kernel_ns = (long long)(fused->level_kernel_ms * 1e6);
      for (int i = 0; i < fused->batches; ++i) last.batch_ms.push_back(fused->batch_ms[i]);
      // the long-row queue only exists on push levels; the short-row queue gets the rest of the push edges
      last.stream.launches = fused->stream_kernel_launches;
      last.stream.ns = (long long)(fused->stream_kernel_ms * 1e6);
      last.wave.launches = fused->wave_kernel_launches;
      last.wave.ns = (long long)(fused->wave_kernel_ms * 1e6);
    }
    if (with_timing && fused->time_kernels == 2) {            // the merged push launch: everything the push levels expanded
      last.stream.edges = last.push_edges;                    // (the "wave" half then holds the queue builds: launches and ns only)
      last.stream.vertices = push_vertices;
    } else if (!direction_optimizing) {
      last.stream.edges = (long long)hc->sum_long_edges;
      last.stream.vertices = (long long)hc->sum_long_vertices;
      last.wave.edges = last.push_edges - last.stream.edges;
      last.wave.vertices = push_vertices - last.stream.vertices;
    }
    last.dominant = (last.stream.ns > last.wave.ns || (with_timing && fused->time_kernels == 2)) ? 1 : 0;
    for (int i = 0; i < 64; ++i) last.claims_level[i] = (long long)hc->claims_level[i];
  }

  // labels are (re)initialised by the run itself; bfs_problem->src is the source.
  // direction_optimizing: bottom-up levels once num_unvisited < frontier_length * alpha
  // (the reference's rule, bfs_enactor.hxx:68); in-edges come from the graph's CSC slots.
  void enact(std::shared_ptr<bfs_problem_t> bfs_problem, standard_context_t& context,
             bool direction_optimizing = false, float alpha = 0.f) {
    auto& g = *bfs_problem->gslice;
    mgx::bfs_layout_t layout = layout_of(g);
    // the hub-first layout carries no separate CSC: bottom-up levels can use it only on graphs whose
    // CSC slots alias the CSR (symmetric input, what the reference always has)
    const bool use_layout = g.has_layout && (!direction_optimizing || g.csc_is_csr);
    last = bfs_run_stats_t();
    // (one source: no wait for its shape -- known: used; not known: asked for in front of the traversal, collected behind it)
    bool asked = false;
    if (use_layout && g.src_shapes_enabled && mgx::bfs_wants_src_shapes(*fused, direction_optimizing ? 1 : 0) && g.num_edges > 0) {
      if (g.src_shape_cache.lookup_or_request(g.d_row_offsets.data(), g.d_col_indices.data(), g.num_nodes, fused->long_min, bfs_problem->src, shape_table, context)) {
        layout.src_shapes = shape_table.data();
        layout.src_shapes_long_min = fused->long_min;
      } else asked = true;
    }
    mgx::bfs_fused_run(*fused, g.d_row_offsets.data(), g.d_col_indices.data(), bfs_problem->d_labels.data(),
                       bfs_problem->src, context, use_layout ? &layout : nullptr, direction_optimizing ? 1 : 0, alpha,
                       g.d_col_offsets.data(), g.d_row_indices.data());
    if (asked) g.src_shape_cache.collect();            // (bfs_fused_run returns behind its last launch: the copy in front of the first has landed)
    fill_stats(last, fused->host_ctrl, direction_optimizing, true);
  }

  // `count` traversals enqueued back to back, one host wait (mgx::bfs_fused_run_many): out[i] = the counters of source
  // srcs[i]; the labels are those of the LAST source when the call returns.  Returns how many traversals had to be run
  // again on their own (they did not finish within the slots the batch gave them).
  // A batch is submitted in chunks of at most MANY_CHUNK sources -- one enqueue, one host wait and one pinned block of heads
  // per chunk, reused -- so that a long source list neither pins count x ~1 KB of host memory nor queues millions of launches
  // behind one wait (a traversal is ~2 * slots + 4 launches).  A chunk of 512 traversals is ~170 ms of device work on
  // RMAT-22: the extra host wait per chunk (~10 us) does not show.
  static constexpr int MANY_CHUNK = 512;
  char* many_heads = nullptr;       // pinned, MANY_CHUNK heads at most
  int many_cap = 0;
  int enact_many(std::shared_ptr<bfs_problem_t> bfs_problem, standard_context_t& context, const int* srcs, int count,
                 std::vector<bfs_run_stats_t>& out, bool direction_optimizing = false, float alpha = 0.f) {
    auto& g = *bfs_problem->gslice;
    mgx::bfs_layout_t layout = layout_of(g);
    const bool use_layout = g.has_layout && (!direction_optimizing || g.csc_is_csr);
    const int want = count < MANY_CHUNK ? count : MANY_CHUNK;
    if (want > many_cap) {
      if (many_heads) (void)hipHostFree(many_heads);
      many_heads = nullptr;
      many_cap = 0;
      MGX_HIP(hipHostMalloc((void**)&many_heads, (size_t)want * mgx::bfs_many_head_bytes(), hipHostMallocDefault));
      many_cap = want;
    }
    out.assign((size_t)(count > 0 ? count : 0), bfs_run_stats_t());
    int reruns = 0;
    for (int first = 0; first < count; first += MANY_CHUNK) {
      const int part = count - first < MANY_CHUNK ? count - first : MANY_CHUNK;
      if (use_layout) resolve_shapes(g, layout, srcs + first, part, direction_optimizing ? 1 : 0, context);
      reruns += mgx::bfs_fused_run_many(*fused, g.d_row_offsets.data(), g.d_col_indices.data(), bfs_problem->d_labels.data(), srcs + first,
                                        part, context, many_heads, use_layout ? &layout : nullptr, direction_optimizing ? 1 : 0,
                                        alpha, g.d_col_offsets.data(), g.d_row_indices.data());
      for (int i = 0; i < part; ++i) fill_stats(out[(size_t)(first + i)], mgx::bfs_many_head(many_heads, i), direction_optimizing, false);
    }
    if (count > 0) { last = out.back(); bfs_problem->src = srcs[count - 1]; }
    return reruns;
  }
  ~bfs_fused_enactor_t() { if (many_heads) (void)hipHostFree(many_heads); }
};

}  // namespace bfs
}  // namespace gunrock
