/*
 * mgx.h -- C-ABI of the MI355X-native frontier traversal engine (libmgx.so).
 *
 * The reference (gunrock/mini) has no FFI: its boundary is a set of C++ function
 * templates parameterised by device functors.  Device functors cannot
 * cross a C ABI, so this header carries the operators PRE-INSTANTIATED for the
 * in-scope functors (BFS, SSSP, PR), the building blocks (scan, load-balanced
 * search, compaction, segmented reduce) and whole-algorithm runs.  The
 * source-compatible template API itself lives in include/gunrock/ (*.hxx).
 *
 * Conventions: every entry point returns an int status (MGX_OK == 0, negative ==
 * error), never calls exit() (the reference does: frontier.hxx:53-59,83-93,
 * graph.hxx:107-110), never throws across the boundary.  Handles are opaque.
 * Host buffers are caller-owned.  `stream` arguments are hipStream_t passed as
 * void* (NULL == legacy default stream, which is what the reference uses).
 * All ids/offsets are 32-bit like the reference (graph.hxx:19-26); counts that
 * can exceed 2^31 are int64_t.
 */
#ifndef MGX_H_
#define MGX_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGX_API __attribute__((visibility("default")))

/* status codes */
#define MGX_OK 0
#define MGX_E_INVALID (-1)           /* bad argument                                         */
#define MGX_E_HIP (-2)               /* a HIP runtime call failed (see mgx_last_error)        */
#define MGX_E_FRONTIER_OVERFLOW (-4) /* frontier.hxx:53-59,83-93: the reference exit(0)s here */
#define MGX_E_NEGATIVE_WEIGHT (-5)   /* SSSP relaxes on the int view of non-negative floats   */
#define MGX_E_NO_DEVICE (-6)

typedef struct mgx_ctx_s* mgx_ctx_t;
typedef struct mgx_graph_s* mgx_graph_t;
typedef struct mgx_frontier_s* mgx_frontier_t;
typedef struct mgx_bfs_s* mgx_bfs_t;
typedef struct mgx_sssp_s* mgx_sssp_t;
typedef struct mgx_pr_s* mgx_pr_t;
typedef struct mgx_kcore_s* mgx_kcore_t;
typedef struct mgx_dbfs_s* mgx_dbfs_t;
typedef struct mgx_dbfs2_s* mgx_dbfs2_t;
typedef struct mgx_dsssp_s* mgx_dsssp_t;
typedef struct mgx_comm_s* mgx_comm_t;

MGX_API int mgx_version(void);
/* The environment switches the library reads (include/mgx/env.hpp: the one table, the one reader): returns how many there are; for
 * 0 <= index < that, *name / *what (either may be NULL) point at the switch's name and a line about it (static strings).  Every switch
 * selects between product paths the thresholds pick by size; none changes a result.  An empty value counts as unset. */
MGX_API int mgx_env_switches(int index, const char** name, const char** what);
MGX_API int mgx_build_is_lab(void); /* 1: built with -DMGX_LAB (experiment shapes and instrumented kernels compiled in: never the product) */
MGX_API const char* mgx_strerror(int status);
MGX_API const char* mgx_last_error(void); /* thread-local detail of the last failure */

/* ---- context: replaces mgpu::standard_context_t (tests/bfs/test_bfs.cu:22) ---- */
MGX_API int mgx_ctx_create(int device, void* stream, mgx_ctx_t* out);
MGX_API int mgx_ctx_set_stream(mgx_ctx_t ctx, void* stream);
MGX_API int mgx_ctx_synchronize(mgx_ctx_t ctx);
MGX_API int mgx_ctx_destroy(mgx_ctx_t ctx);
MGX_API int mgx_ctx_num_cus(mgx_ctx_t ctx, int* out);

/* ---- graph: replaces graph_device_t + graph_to_device (graph.hxx:37-83) ----
 * col_offsets/row_indices/row_weights == NULL mirrors the CSR into the "CSC" slots, which
 * is what the reference always ends up doing.  weights == NULL means all 1.0f
 * (graph.hxx:126).  upload copies from host; wrap_device borrows device pointers (e.g.
 * torch tensors) that must outlive the graph.                                           */
MGX_API int mgx_graph_upload(mgx_ctx_t ctx, int num_nodes, int64_t num_edges,
                             const int* row_offsets, const int* col_indices, const float* weights,
                             const int* col_offsets, const int* row_indices, const float* row_weights,
                             mgx_graph_t* out);
MGX_API int mgx_graph_wrap_device(mgx_ctx_t ctx, int num_nodes, int64_t num_edges,
                                  const int* d_row_offsets, const int* d_col_indices, const float* d_weights,
                                  const int* d_col_offsets, const int* d_row_indices, const float* d_row_weights,
                                  mgx_graph_t* out);
/* Optional hub-first layout used by mgx_bfs_run (not in the reference): the same graph with vertex
 * ids renumbered by DESCENDING DEGREE (layout id 0 = highest degree), as CSR, plus the two id maps
 * (new_of_old[old] = layout id, old_of_new = inverse).  With it the hot prefix of the visited
 * bitmap (the vertices that receive most edges) is kept in LDS.  Device pointers are borrowed.
 * Results (labels) are always in original ids; the operator entry points ignore the layout.   */
MGX_API int mgx_graph_attach_layout(mgx_graph_t g, const int* d_layout_row_offsets, const int* d_layout_col_indices,
                                    const int* d_new_of_old, const int* d_old_of_new);
/* weights of the layout's edges, in the layout's order (optional): mgx_sssp_run then relaxes in layout space, where
 * the distances of the high-degree vertices -- the targets of most relaxations -- sit together in L2 */
MGX_API int mgx_graph_attach_layout_weights(mgx_graph_t g, const float* d_layout_weights);
/* The same layout built by the library from the graph's own CSR (device-side degree sort, renumbering, rows sorted by
 * neighbour; one-time setup, rocPRIM sort / scan) and owned by the graph; with_weights != 0 carries the edge weights
 * along for the fused SSSP loop.  mgx_graph_layout_read copies the pieces to host buffers (NULL: skip; sizes n + 1,
 * m, n, n, m) -- what the tests compare with the torch construction in mini_amd/rmat.py. */
MGX_API int mgx_graph_build_layout(mgx_graph_t g, int with_weights);
/* What the layout holds: out8 = { 1 if the graph has one, unit blocks (64-entry units of the long rows), 1 if their 24-bit copy
 * exists, pairs of the cold-edge lists, slices that hold pairs, units of the blocks WITHOUT the lists' entries (the fused BFS reads
 * those when it runs the cold-edge pass), 1 if the long rows' entries were mostly cold (no lists: a flat graph), bytes of device
 * memory the library allocated for the layout and everything cut from it (arrays the caller attached are not counted) }. */
MGX_API int mgx_graph_layout_info(mgx_graph_t g, int64_t* out8);
/* The long rows regrouped by slice of their destinations for the full-frontier neighbour-reduce (mgx/nreduce.hpp: k_nrs_edges; replaces
 * the gather of value(u) per edge of the reference's neighborhood_kernel, neighborhood.hxx:39-58, by LDS reads): built by the library
 * at the graph's first full-frontier reduce (mgx_segreduce_*, mgx_pr_enact) on a graph with the library's layout.  out5 = { 16-byte
 * mini-units (0: not built), hot slices of 40 000 vertices, long rows, rows whose partials more than one lane folds, mini-units of the tail }. */
MGX_API int mgx_graph_nr_slices_info(mgx_graph_t g, int64_t* out5);
MGX_API int mgx_graph_layout_read(mgx_graph_t g, int* h_row_offsets, int* h_col_indices, int* h_new_of_old,
                                  int* h_old_of_new, float* h_weights);
/* Genuine CSC (the transpose) built by the library from the graph's own CSR, device-side (one stable radix sort of the
 * edges by destination), owned by the graph: col_offsets / row_indices (sources of every vertex's in-edges, ascending) /
 * row_values.  What bottom-up BFS levels need on DIRECTED inputs (BASELINE config 4); the reference's load_graph means to
 * build it (graph.hxx:176-222) but returns the CSR in the CSC slots.  mgx_graph_csc_read copies the CSC slots
 * to host buffers (NULL: skip; sizes n + 1, m, m) -- whatever they hold, alias or transpose. */
MGX_API int mgx_graph_build_csc(mgx_graph_t g);
MGX_API int mgx_graph_csc_read(mgx_graph_t g, int* h_col_offsets, int* h_row_indices, float* h_row_values);
MGX_API int mgx_graph_free(mgx_graph_t g);
MGX_API int mgx_graph_dims(mgx_graph_t g, int* num_nodes, int64_t* num_edges);
/* host-side MTX text loader, bug-compatible with load_graph (graph.hxx:96-223): row = 2nd
 * field, neighbour = 1st field, multi-edges and self loops kept, undir appends the swapped
 * copies.  Returns malloc'd arrays to be released with mgx_host_free.                    */
MGX_API int mgx_load_mtx(const char* path, int undir, int random_edge_value,
                         int* num_nodes, int64_t* num_edges,
                         int** row_offsets, int** col_indices, float** weights);
/* The same plus the loader's CSC slots: the CSR again (what the reference always returns) or, with
 * genuine_csc != 0 and undir == 0, the transpose (load_graph's _genuine_csc option: what graph.hxx:176-222 means to
 * build).  Six malloc'd arrays to be released with mgx_host_free.                                  */
MGX_API int mgx_load_mtx_csc(const char* path, int undir, int random_edge_value, int genuine_csc,
                             int* num_nodes, int64_t* num_edges, int** row_offsets, int** col_indices, float** weights,
                             int** col_offsets, int** row_indices, float** row_weights);
/* Binary CSR cache: the loader's output as raw arrays behind a header with a checksum, so that a big
 * MatrixMarket file is parsed and sorted once (mgx_load_mtx / mgx_load_mtx_csc, then mgx_graph_save_csr) and mapped
 * back without parsing afterwards (mgx_graph_load_csr).  col_offsets / row_indices / row_weights: a genuine CSC to store
 * with it, or all NULL.  load: a file that is missing, truncated, of another format or fails its checksum or the
 * structural checks (monotone offsets, ids in range) is MGX_E_INVALID -- a cache is never trusted; the CSC outputs are
 * NULL when the file holds none.  Arrays are malloc'd: release with mgx_host_free. */
MGX_API int mgx_graph_save_csr(const char* path, int num_nodes, int64_t num_edges, int undirected, const int* row_offsets,
                               const int* col_indices, const float* weights, const int* col_offsets, const int* row_indices,
                               const float* row_weights);
MGX_API int mgx_graph_load_csr(const char* path, int* num_nodes, int64_t* num_edges, int* undirected, int** row_offsets,
                               int** col_indices, float** weights, int** col_offsets, int** row_indices, float** row_weights);
MGX_API void mgx_host_free(void* p);

/* ---- frontier: replaces frontier_t<int> (frontier.hxx:12-99) ---- */
MGX_API int mgx_frontier_create(mgx_ctx_t ctx, int64_t capacity, mgx_frontier_t* out);
MGX_API int mgx_frontier_free(mgx_frontier_t f);
MGX_API int mgx_frontier_load(mgx_frontier_t f, const int* host, int64_t n);   /* load(vector)  :65-79 */
MGX_API int mgx_frontier_fill_iota(mgx_frontier_t f, int64_t n);              /* enactor.hxx:29-34   */
MGX_API int mgx_frontier_fill(mgx_frontier_t f, int value, int64_t n);        /* fill + load(mem_t)  */
MGX_API int mgx_frontier_read(mgx_frontier_t f, int* host, int64_t cap, int64_t* n);
MGX_API int mgx_frontier_resize(mgx_frontier_t f, int64_t n);                 /* resize :82-93       */
MGX_API int mgx_frontier_size(mgx_frontier_t f, int64_t* n);
MGX_API int mgx_frontier_capacity(mgx_frontier_t f, int64_t* n);
MGX_API int mgx_frontier_device_ptr(mgx_frontier_t f, int** d_ptr);

/* ---- building blocks (moderngpu call sites: advance.hxx:40,62; filter.hxx:18-29;
 *      neighborhood.hxx:35,58).  Device pointers in, counts out.                       */
/* out[i] = sum_{j<i} in[j]; *total = sum of all (transform_scan<int>, plus_t)          */
MGX_API int mgx_scan_exclusive_i32(mgx_ctx_t ctx, const int* d_in, int64_t n, int* d_out, int64_t* total);
/* degree scan of a frontier over the graph's row offsets (push) or col offsets (pull):
 * d_scanned_row_offsets[i] = sum_{j<i} deg(in[j])   (advance.hxx:32-43)                 */
MGX_API int mgx_scan_frontier_degrees(mgx_graph_t g, mgx_frontier_t in, int use_csc, int64_t* total);
/* transform_lbs made visible: (seg, rank) of every work item of the last degree scan  */
MGX_API int mgx_lbs_expand_debug(mgx_graph_t g, mgx_frontier_t in, int64_t total,
                                 int* host_seg, int* host_rank);
/* stable compaction of d_in by (d_in[i] != drop_value) -- transform_compact            */
MGX_API int mgx_compact_i32(mgx_ctx_t ctx, const int* d_in, int64_t n, int drop_value, int* d_out, int64_t* kept);
/* neighborhood_kernel (neighborhood.hxx:12-70) with get_value_to_reduce = d_vertex_value[nbr]:
 * d_reduced[seg] = op over the segment's neighbours, identity for empty segments.
 * push != 0 walks the CSR, 0 the CSC slots.                                             */
MGX_API int mgx_segreduce_f32_plus(mgx_graph_t g, mgx_frontier_t in, int push,
                                   const float* d_vertex_value, float identity, float* d_reduced, int64_t* nonzeros);
MGX_API int mgx_segreduce_i32_min(mgx_graph_t g, mgx_frontier_t in, int push,
                                  const int* d_vertex_value, int identity, int* d_reduced, int64_t* nonzeros);
MGX_API int mgx_segreduce_i32_max(mgx_graph_t g, mgx_frontier_t in, int push,
                                  const int* d_vertex_value, int identity, int* d_reduced, int64_t* nonzeros);

/* ---- BFS: bfs_problem_t / bfs_functor_t / bfs_enactor_t (gunrock/src/bfs/) ---- */
MGX_API int mgx_bfs_create(mgx_graph_t g, int src, mgx_bfs_t* out);       /* bfs_problem.hxx:34-46 */
MGX_API int mgx_bfs_reset(mgx_bfs_t p, int src);
MGX_API int mgx_bfs_free(mgx_bfs_t p);
MGX_API int mgx_bfs_labels(mgx_bfs_t p, int* host_labels);                /* extract() :48-50      */
MGX_API int mgx_bfs_preds(mgx_bfs_t p, int* host_preds);                  /* stay -1   */
MGX_API int mgx_bfs_labels_device(mgx_bfs_t p, int** d_labels);
/* advance_forward_kernel<bfs_problem_t,bfs_functor_t,false,true>  (bfs_enactor.hxx:52-57) */
MGX_API int mgx_bfs_advance(mgx_bfs_t p, mgx_frontier_t in, mgx_frontier_t out, int iteration, int64_t* front);
/* filter_kernel<bfs_problem_t,bfs_functor_t>                      (bfs_enactor.hxx:60-65) */
MGX_API int mgx_bfs_filter(mgx_bfs_t p, mgx_frontier_t in, mgx_frontier_t out, int iteration, int64_t* kept);
/* the two above in one pass over the edges: out = ids whose label this call set          */
MGX_API int mgx_bfs_advance_filter_fused(mgx_bfs_t p, mgx_frontier_t in, mgx_frontier_t out, int iteration, int64_t* kept);
/* The reference's IDEMPOTENT mode (no upstream enactor uses it):
 * advance_forward_kernel<bfs_problem_t, F, idempotence = true, true> (advance.hxx:60): out = EVERY neighbour of the
 * frontier, duplicates and visited vertices included, no atomics on labels; front = edges expanded.
 * uniquify_kernel<bfs_problem_t, F> (filter.hxx:95-119): out = the vertices of `in` not seen before in this traversal,
 * one copy each, stable (wave64 intra-wave cull, then an exact visited-bitmask cull: the mask, (n + 31) / 32 words,
 * belongs to the problem handle and is reset by mgx_bfs_reset with the source's bit set), labelled iteration + 1 by
 * F::cond_uniq.  F = bfs_idempotent_functor_t (include/gunrock/bfs/bfs_functor.hxx).
 * mgx_bfs_enact_idempotent: the superstep loop over the two from the problem's source (labels as mgx_bfs_reset left
 * them); stats[0] = iterations, [1] = edges expanded.  Labels equal enact_pushpull's. */
MGX_API int mgx_bfs_advance_idempotent(mgx_bfs_t p, mgx_frontier_t in, mgx_frontier_t out, int iteration, int64_t* front);
MGX_API int mgx_bfs_uniquify(mgx_bfs_t p, mgx_frontier_t in, mgx_frontier_t out, int iteration, int64_t* kept);
MGX_API int mgx_bfs_enact_idempotent(mgx_bfs_t p, int64_t* stats);
/* gen_unvisited_kernel / sparse_to_dense_kernel / advance_backward_kernel (advance.hxx:69-160) */
MGX_API int mgx_bfs_gen_unvisited(mgx_bfs_t p, mgx_frontier_t indices, mgx_frontier_t unvisited, int iteration, int64_t* kept);
MGX_API int mgx_bfs_sparse_to_dense(mgx_bfs_t p, mgx_frontier_t sparse, mgx_frontier_t dense, int iteration);
MGX_API int mgx_bfs_advance_backward(mgx_bfs_t p, mgx_frontier_t unvisited, mgx_frontier_t bitmap,
                                     mgx_frontier_t bitmap_out, int iteration, int64_t* front);
/* bfs_enactor_t::enact_pushpull (bfs_enactor.hxx:41-117) on the operator-per-superstep path.
 * stats[0]=pushed iterations, [1]=total iterations, [2]=edges expanded by push,
 * [3]=in-edges inspected by pull.                                                        */
MGX_API int mgx_bfs_enact_pushpull(mgx_bfs_t p, float threshold, int64_t* stats);

/* Whole traversal on the fused device-resident path (no per-level host round trip):
 * labels identical to enact_pushpull's.  mode: MGX_BFS_PUSH (config 2) or
 * MGX_BFS_DIRECTION_OPT (config 4; needs a genuine CSC for directed graphs).
 * A traversal is enqueued as batches of launch slots (mgx/bfs_fused_run.hpp); the host waits once per batch
 * (first batch: as many slots as the previous traversal of the graph needed, then MGX_BFS_LEVELS_PER_SYNC).
 * stats (may be NULL):
 *   [0] levels run  [1] vertices reached (incl. source)  [2] m_t = sum of out-degrees of
 *   reached vertices (the TEPS numerator)  [3] edges inspected by push levels
 *   [4] edges inspected by pull levels [5] push levels [6] level-kernel launches (incl. the
 *   empty ones behind the last level) [7] device time of those launches in ns (HIP events on
 *   the context's stream) [8] frontier vertices expanded (sum of per-level frontier sizes)
 *   [9] visited-bit claims (atomicOr) issued.
 *   Timed push kernel (only with mgx_bfs_set_kernel_timing; mode 2: the merged k_bfs_push launch, mode 1: the part
 *   with more device time in this run):
 *   [10] its launches (incl. the ones that find nothing to do) [11] their device time in ns (HIP events
 *   around every launch) [12] edges and [13] frontier vertices it processed
 *   [14] which one (1 = long-row part or the merged launch, 0 = short-row part)
 *   [15] levels run inside a push launch's block 0 (chains of small levels, bfs_fused_chain.hpp)
 *   [16] launch slots used [17] slots whose long rows were read from the unit blocks (bfs_fused_dense.hpp)
 *   [18] slots whose short rows were walked vertex by vertex (bfs_fused_vshort.hpp)
 *   [19] slots that ran without queues (the build before them wrote none, bfs_build_is_lazy)
 *   [20] slots that ran the cold-edge pass (bfs_fused_cold.hpp) [21] mid-size levels expanded by M launches (bfs_fused_mini.hpp).
 *   mgx_bfs_run writes entries [0] .. [15] (stats must hold 16: the contract of the first release, kept so that a
 *   caller built against it is not overrun); mgx_bfs_run_stats is the same call with an explicit capacity and writes
 *   min(cap, 24) entries.                                                                       */
#define MGX_BFS_PUSH 0
#define MGX_BFS_DIRECTION_OPT 1
MGX_API int mgx_bfs_run(mgx_bfs_t p, int src, int mode, float alpha, int64_t* stats);
MGX_API int mgx_bfs_run_stats(mgx_bfs_t p, int src, int mode, float alpha, int64_t* stats, int cap);
/* A batch of sources: `count` complete traversals (each as mgx_bfs_run: labels re-initialised, every level run, counters
 * taken) enqueued back to back on the context's stream with ONE host wait at the end -- the reference's one-call-per-
 * source driver (test_bfs.cu:32-42) pays a host round trip per traversal, ~18 us of a 0.35 ms RMAT-22 traversal here.
 * stats: count x cap entries, row i = the counters of sources[i] (the layout of mgx_bfs_run_stats; timing entries are 0);
 * the labels (mgx_bfs_labels) are those of the LAST source.  *reruns (optional): traversals that did not finish inside the
 * batch and were run again on their own (a source whose level structure needed more launch slots than its predecessors). */
MGX_API int mgx_bfs_run_many(mgx_bfs_t p, const int* sources, int count, int mode, float alpha, int64_t* stats, int cap, int* reruns);
/* per-level trace of the last mgx_bfs_run: level_nf[i], level_edges[i] for i < *levels  */
MGX_API int mgx_bfs_level_trace(mgx_bfs_t p, int cap, int64_t* level_nf, int64_t* level_edges, int* levels);

/* device time (ms, HIP events) of each launch batch of the last mgx_bfs_run; with the environment
 * variable MGX_BFS_LEVELS_PER_SYNC=1 a batch is exactly one level kernel                   */
/* Per-launch timing of the push kernels is OFF by default: every hipEventRecord between two kernels leaves a
 * ~6 us gap on the stream (measured, rocprofv3 kernel trace).  on == 1: following mgx_bfs_run calls launch the parts of
 * a slot's push (long rows, short rows) separately with events around each and fill mgx_bfs_kernel_times /
 * mgx_bfs_level_kernel_times; on == 2: events around the ONE merged push launch of every slot (k_bfs_push, the kernel a
 * traversal actually runs): its launches and time are reported in the "stream" half of mgx_bfs_kernel_times with ALL
 * push edges and frontier vertices; the "wave" half then holds the launches and time of the slots' queue builds
 * (k_bfs_build2 -- the filter half of advance + filter), edges and vertices zero. */
MGX_API int mgx_bfs_set_kernel_timing(mgx_bfs_t p, int on);
/* the two push kernels of the last mgx_bfs_run, timed per launch with HIP events on the context's stream:
 * out8 = { stream launches, ns, edges, frontier vertices,  wave launches, ns, edges, frontier vertices }
 * ("stream": the long-row part of k_bfs_push -- rows of >= MGX_BFS_LONG_MIN edges, unit blocks or queue walk, with the
 *  cold-edge pass -- or, in timing mode 2, the whole merged launch; "wave": the short-row part)               */
MGX_API int mgx_bfs_kernel_times(mgx_bfs_t p, int64_t* out8);
/* duration of each level of the last run (ms), from device-side timestamps taken when a level is opened: no host
 * synchronisation involved; first min(cap, 63) levels */
MGX_API int mgx_bfs_level_times(mgx_bfs_t p, int cap, float* ms, int* levels);
/* the same per launch slot (ms), first min(cap, 64) slots of the last run; a slot = one k_bfs_push launch
 * and its queue build (include/mgx/bfs_fused_run.hpp) */
MGX_API int mgx_bfs_level_kernel_times(mgx_bfs_t p, int cap, float* stream_ms, float* wave_ms);
/* atomicOr claims issued per level of the last run (first 64 levels) */
MGX_API int mgx_bfs_level_claims(mgx_bfs_t p, int cap, int64_t* claims);
MGX_API int mgx_bfs_batch_times(mgx_bfs_t p, int cap, float* ms, int* batches);

/* ---- vertex-range partitioned BFS: the per-rank pieces (the reference has no
 *      multi-GPU path, README.md:4).  Rank r of `ranks` owns global ids [r*chunk, (r+1)*chunk),
 *      chunk = ceil(n_global/ranks): their CSR rows (local row_offsets, GLOBAL col_indices) and
 *      labels.  Per superstep the host calls expand (-> per-owner bins of neighbour ids, each id at
 *      most once per traversal per rank), exchanges bin r of every rank to rank r (all-to-all over
 *      RCCL/xGMI), calls receive for what arrived (and for its own bin) and swap.  Labels equal the
 *      single-GPU depths bit for bit.  Device pointers are borrowed.                          */
/* d_bins: optional caller-owned send buffer of ranks*bin_capacity ints (bin_capacity >= chunk), so the
 * host can hand slices of it straight to its collective; NULL = library-owned.                  */
MGX_API int mgx_dbfs_create(mgx_ctx_t ctx, int n_global, int ranks, int rank, int64_t m_local,
                            const int* d_row_offsets_local, const int* d_col_indices_global,
                            int* d_bins, int64_t bin_capacity, mgx_dbfs_t* out);
MGX_API int mgx_dbfs_free(mgx_dbfs_t h);
MGX_API int mgx_dbfs_range(mgx_dbfs_t h, int* v_lo, int* v_hi);
MGX_API int mgx_dbfs_reset(mgx_dbfs_t h, int src_global);
MGX_API int mgx_dbfs_expand(mgx_dbfs_t h, int64_t* host_counts_per_rank, int64_t* edges_expanded);
MGX_API int mgx_dbfs_bins(mgx_dbfs_t h, int** d_bins, int64_t* bin_capacity); /* bin r at d_bins + r*capacity */
MGX_API int mgx_dbfs_receive(mgx_dbfs_t h, const int* d_global_ids, int64_t count, int label);
MGX_API int mgx_dbfs_swap(mgx_dbfs_t h, int64_t* next_frontier_size);
MGX_API int mgx_dbfs_labels(mgx_dbfs_t h, int* host_labels_local);

/* ---- vertex-range partitioned SSSP: the per-rank pieces ("(dst, dist) pairs with min-combining"; the
 *      reference has no multi-GPU path).  Same partition as mgx_dbfs_*: rank r owns [r*chunk, (r+1)*chunk), their rows
 *      (local row_offsets, GLOBAL col_indices, weights) and distances.  A superstep is frontier Bellman-Ford
 *      (sssp_enactor.hxx:40-72): expand relaxes the local frontier's edges -- local targets at once, remote ones into
 *      per-owner bins of 8-byte pairs (vertex << 32 | float bits), ONE pair per target vertex and superstep carrying the
 *      minimum over all of this rank's edges to it (min-combining before send) and only if it beats everything this
 *      rank sent for that vertex before; the host exchanges bin r of every rank to rank r (all-to-all-v); receive keeps
 *      the minimum and queues the vertices whose distance dropped; swap returns the next local frontier size (the host
 *      all-reduces it: zero everywhere ends the run).  Distances equal the single-GPU ones bit for bit; unreachable
 *      vertices hold FLT_MAX.  Device pointers are borrowed. */
MGX_API int mgx_dsssp_create(mgx_ctx_t ctx, int n_global, int ranks, int rank, int64_t m_local, const int* d_row_offsets_local,
                             const int* d_col_indices_global, const float* d_weights, mgx_dsssp_t* out);
MGX_API int mgx_dsssp_free(mgx_dsssp_t h);
MGX_API int mgx_dsssp_reset(mgx_dsssp_t h, int src_global);
MGX_API int mgx_dsssp_expand(mgx_dsssp_t h, int64_t* host_counts_per_rank, int64_t* edges_relaxed);
MGX_API int mgx_dsssp_bins(mgx_dsssp_t h, uint64_t** d_bins, int64_t* bin_capacity); /* bin r at d_bins + r*capacity */
MGX_API int mgx_dsssp_receive(mgx_dsssp_t h, const uint64_t* d_pairs, int64_t count);
MGX_API int mgx_dsssp_swap(mgx_dsssp_t h, int64_t* next_frontier_size);
MGX_API int mgx_dsssp_distances(mgx_dsssp_t h, float* host_dist_local);
/* The whole superstep loop from `src_global` with the library's communicator (mgx_comm_create; NULL for one rank): expand,
 * bin counts by all-gather, the all-to-all-v of the pairs as one group of sends and receives, receive, swap, global
 * frontier size -- until it is zero.  out4: supersteps, edges relaxed here, pairs sent from here, pairs received here.
 * Every rank of the communicator must make the call.  No reference counterpart (README.md:4); SURVEY 8e. */
MGX_API int mgx_dsssp_run(mgx_dsssp_t h, mgx_comm_t comm, int src_global, int64_t* out4);

/* ---- partitioned BFS, generation 2 (include/mgx/bfs_dist2.hpp): every rank runs the FUSED level
 *      kernels on its rows and ranks exchange dense "newly visited" bitmaps (one all-gather of n/8 bytes
 *      per rank and level) instead of id lists.  Ids are global and hub-first (descending global degree);
 *      vertex v belongs to rank v % ranks, local row v / ranks.  d_newbits: caller-owned buffer of
 *      mgx_dbfs2_words(n_global) words that mgx_dbfs2_push fills with the rank's discoveries of the level;
 *      mgx_dbfs2_merge takes the all-gathered ranks x words array, ORs it into every rank's visited
 *      bitmap, labels the owned vertices and builds the rank's next frontier (returns its size/edges). */
MGX_API int mgx_dbfs2_create(mgx_ctx_t ctx, int n_global, int ranks, int rank, const int* d_row_offsets_local,
                             const int* d_col_indices_global, unsigned* d_newbits, mgx_dbfs2_t* out);
/* Unit blocks of the rank's long rows (64-entry units, one owner each, as mgx_graph_build_layout makes them for the
 * single-GPU path): levels that hold a large share of the rank's long rows are then read from them, 16 bytes per lane,
 * with the level's merged discoveries as the frontier bitmap, instead of walking the long-row queue.  On a graph that
 * outgrows the LDS prefix of the visited bitmap (652 288 vertices) the call also builds the rows' cold-edge lists -- the
 * entries behind the prefix as (owner, destination) pairs by slice of the destination: workgroups of the push launch take
 * them with THEIR slice of the bitmap in LDS and leave a bitmap, no byte marks (needs rows sorted by neighbour id, as the
 * library's shard builder makes them).  Once per engine, optional; *units (may be NULL) <- units built (0: the rows are
 * all short).  Needs row_offsets / col_indices to stay valid and unchanged.  No reference counterpart.
 * Round 4: with cold-edge lists the unit blocks hold the rows' entries INSIDE the prefix only, three bytes each, and the pairs
 * are kept a second time at four bytes each; when the rank's rows come by non-increasing degree (the shard builder's order) the
 * call also prepares the vertex-by-vertex walk of its short rows -- a copy of col_indices with readable entries behind it
 * (m_local + 8 ints of device memory) and a bitmap of the rank's local frontier.  mgx_dbfs2_path_levels says what a traversal used. */
MGX_API int mgx_dbfs2_build_units(mgx_dbfs2_t h, int64_t* units);
/* levels of the traversal whose long rows were read from the unit blocks, as of the last mgx_dbfs2_status / mgx_dbfs2_run */
MGX_API int mgx_dbfs2_dense_levels(mgx_dbfs2_t h, int64_t* levels);
/* ... and how many of them ran the cold-edge pass (the long rows' entries behind the LDS prefix as (owner, destination) pairs by
 * slice of the destination, built with the unit blocks when the graph is big enough to have any: *pairs, may be NULL) */
MGX_API int mgx_dbfs2_cold_levels(mgx_dbfs2_t h, int64_t* levels, int64_t* pairs);
/* which of round 4's paths the last traversal took on this rank (as of the last mgx_dbfs2_status / mgx_dbfs2_run): out4 = { levels
 * whose push appended its discoveries to the id list itself (no sweep over the marks), levels whose short rows were walked vertex
 * by vertex, 1 if a sweep declared its list overflowed from the push's mark count, slices of the cold-edge lists stored at four
 * bytes per pair }.  Tests and tools. */
MGX_API int mgx_dbfs2_path_levels(mgx_dbfs2_t h, int64_t* out4);
MGX_API int mgx_dbfs2_free(mgx_dbfs2_t h);
/* All of reset / push / merge are asynchronous on the context's stream: a level is
 *   push(level) -> all-gather of d_newbits into d_gathered (the caller's collective, stream-ordered) -> merge(level)
 * and several levels can be enqueued before mgx_dbfs2_status synchronises.  Levels enqueued after the traversal has
 * ended are no-ops on every rank (their new-bit maps are empty).  Bitmaps are mgx_dbfs2_words(n) 32-bit words long
 * (padded to 16 bytes). */
MGX_API int mgx_dbfs2_words(int n_global, int64_t* words);
MGX_API int mgx_dbfs2_reset(mgx_dbfs2_t h, int src_global);
MGX_API int mgx_dbfs2_push(mgx_dbfs2_t h, int level);
MGX_API int mgx_dbfs2_merge(mgx_dbfs2_t h, int level, const unsigned* d_gathered);
/* The same for `maps` new-bit maps that lie `stride_words` apart (a multiple of 4, >= mgx_dbfs2_words): the padded
 * buffers of an all-gather, or ONE map that already is the OR of every rank's -- what a caller has after a
 * reduce-scatter (all-to-all of slices + OR) followed by an all-gather of the merged slices, which moves
 * 2 (R-1)/R bitmaps per rank and level instead of R-1. */
MGX_API int mgx_dbfs2_merge_maps(mgx_dbfs2_t h, int level, const unsigned* d_maps, int maps, int64_t stride_words);
/* The reduce step of that exchange: d_out[w] = OR over r < maps of d_maps[r * stride_words + w], w < words (words and
 * stride multiples of 4; d_out may be d_maps itself).  Asynchronous on the context's stream. */
MGX_API int mgx_dbfs2_or_maps(mgx_dbfs2_t h, const unsigned* d_maps, int maps, int64_t stride_words, int64_t words,
                              unsigned* d_out);
/* Synchronises.  next_level = number of levels enqueued so far.  out6: [0] traversal over (a level discovered nothing
 * on ANY rank -- the same on every rank, no reduction needed) [1] levels that hold vertices [2] edges this rank has
 * expanded [3] vertices discovered by all ranks in the level merged last [4] size and [5] edges of this rank's next
 * queues */
/* Sparse levels (lists on sparse levels, bitmaps on dense ones).  With a list set (mgx_dbfs2_set_list: a device
 * buffer of mgx_dbfs2_list_words(n, ranks) words -- 4 header words, [0] = count, then ids), mgx_dbfs2_push also writes the
 * rank's discoveries of the level there as vertex ids; a count above the capacity means "did not fit".  The caller all-gathers
 * the lists and hands them to mgx_dbfs2_apply_lists, which answers out3 = { 1: some list overflowed, NOTHING was applied,
 * exchange the bitmaps and call mgx_dbfs2_merge* as before | 0: the level is merged (every listed vertex decided once on
 * every rank, the owner's labels and next queues written); sum of the counts (0: no rank discovered anything -- the
 * traversal is over); 0 }.  It waits for that verdict only (a spin on pinned memory), not for the stream.             */
/* A rank's shard of the symmetrised R-MAT graph (scale, edgefactor, seed) under this layout, built inside the library on
 * the rank's GPU with no communication (every rank derives the same hub-first permutation from the same counter-based pair
 * stream): mgx_dbfs2_shard_plan makes the permutation and says how many rows and entries the rank holds; the caller
 * allocates; mgx_dbfs2_shard_fill writes local row offsets (n_local + 1), global neighbour ids (m_local) and, where not
 * NULL, new_of_old / old_of_new / degree_of_new (n each); mgx_dbfs2_shard_free releases the plan.  Untimed setup.     */
MGX_API int mgx_dbfs2_shard_plan(mgx_ctx_t ctx, int scale, int edgefactor, uint64_t seed, int ranks, int rank, void** plan, int* n_local, int64_t* m_local);
MGX_API int mgx_dbfs2_shard_fill(mgx_ctx_t ctx, void* plan, int* d_row_offsets_local, int* d_col_indices, int* d_new_of_old, int* d_old_of_new, int* d_degree_of_new);
MGX_API int mgx_dbfs2_shard_free(void* plan);
MGX_API int mgx_dbfs2_list_words(int n_global, int ranks, int64_t* words);
MGX_API int mgx_dbfs2_set_list(mgx_dbfs2_t h, unsigned* d_list, int64_t words);
MGX_API int mgx_dbfs2_apply_lists(mgx_dbfs2_t h, int level, const unsigned* d_lists, int lists, int64_t stride_words, int64_t* out3);
MGX_API int mgx_dbfs2_status(mgx_dbfs2_t h, int next_level, int64_t* out6);
MGX_API int mgx_dbfs2_labels(mgx_dbfs2_t h, int* host_labels_local);
/* this rank's copy of the visited bitmap over ALL vertices (mgx_dbfs2_words words, bit v = vertex v in hub-first global ids):
 * after a traversal every rank holds the same one -- what the tests check */
MGX_API int mgx_dbfs2_visited(mgx_dbfs2_t h, unsigned* host_words);

/* ---- the partitioned traversal driven from C++ over RCCL (include/mgx/comm.hpp, bfs_dist2.hpp: d2_run) ----
 * A communicator of the library's own: rank 0 calls mgx_comm_unique_id, the 128 bytes travel to the other ranks by any
 * means (the Python layer broadcasts them over torch.distributed), every rank calls mgx_comm_create (collective:
 * ncclCommInitRank).  RCCL is resolved at run time from the copy already in the process (PyTorch's librccl.so.1) or the
 * system's; mgx_comm_library says which.  mgx_dbfs2_run then runs a WHOLE traversal: reset, and per level push -> the
 * exchange of the new-bit maps (0: one ncclAllGather; 1: grouped ncclSend/ncclRecv of slices + OR + ncclAllGather of the
 * merged slices) -> merge, enqueued for a batch of levels at a time on the context's stream, one synchronisation per
 * batch; no host code runs between levels.  exchange_words: length of the exchanged map, >= mgx_dbfs2_words(n) and a
 * multiple of 4 * ranks -- the d_newbits buffer given to mgx_dbfs2_create must be that long.  comm may be NULL for
 * ranks == 1.  out6 as mgx_dbfs2_status. */
MGX_API int mgx_comm_unique_id(unsigned char* out128);
MGX_API int mgx_comm_create(mgx_ctx_t ctx, int ranks, int rank, const unsigned char* id128, mgx_comm_t* out);
MGX_API int mgx_comm_free(mgx_comm_t comm);
MGX_API const char* mgx_comm_library(void);
MGX_API int mgx_comm_available(void); /* 1: RCCL is in the process or could be loaded with every entry point the library needs */
/* The in-process stand-in for RCCL (include/mgx/comm_loopback.hpp): G host THREADS of one process as G ranks -- any devices, the
 * tests use one GPU, where RCCL itself refuses a second rank -- so that mgx_dbfs2_run / mgx_dsssp_run themselves can be run and
 * checked with 2 .. 64 ranks on a one-GPU box.  Since round 6 it is a library of its own, mini_amd/libmgx_loopback.so (built by
 * __graft_entry__.build()): libmgx.so only recognises an id that names it and loads the stand-in from next to itself then; without
 * that file both calls below return MGX_E_INVALID.  mgx_comm_loopback_id makes such an id; mgx_comm_create on it blocks until `ranks` threads have joined (or
 * MGX_LOOPBACK_TIMEOUT_S seconds, default 120, have passed: MGX_E_HIP, as every later call on that communicator).  Collectives
 * are host-synchronous device copies with the group semantics of ncclGroupStart / End.  Test infrastructure for the multi-rank
 * loops: nothing selects it by default.  mgx_comm_info: *is_loopback, and the collective rounds its world has completed. */
MGX_API int mgx_comm_loopback_id(unsigned char* out128);
MGX_API int mgx_comm_info(mgx_comm_t comm, int* is_loopback, int64_t* rounds);
/* A communicator's pre-flight, through the very function table the loops use (collective: every rank calls it): all-gather of the
 * first `words` words of d_send into d_gathered (ranks * words), then the slice exchange's pattern -- grouped send of slice r of
 * d_send (ranks * words) to rank r / receive from rank r into slice r of d_alltoall -- and a wait for the stream. */
MGX_API int mgx_comm_selftest(mgx_comm_t comm, const unsigned* d_send, unsigned* d_gathered, unsigned* d_alltoall, int64_t words);
MGX_API int mgx_dbfs2_run(mgx_dbfs2_t h, mgx_comm_t comm, int src_global, int exchange, int64_t exchange_words, int64_t* out6);
/* With id lists, mgx_dbfs2_run enqueues a WHOLE traversal ahead -- per level the push and either the lists or the bitmaps, as the
 * engine's last traversals went (the first one, and MGX_DIST_SPEC=0, look once per level) -- and waits once.  A list that overflows
 * against the plan freezes the traversal at that level on every rank alike; the host then sends that level through the bitmaps
 * and goes on level by level.  out5 = { traversals planned ahead, of those frozen, of those longer than planned, levels and
 * lists-mask (bit L: level L from id lists) of the last plan }. */
MGX_API int mgx_dbfs2_spec_stats(mgx_dbfs2_t h, int64_t* out5);
/* Level 0 of every traversal runs with a host look on every rank, and its all-gathered id-list headers carry each rank's plan: the
 * rest is enqueued ahead only when ALL ranks announced the same plan (round 6) -- a rank whose history differs (its engine handle was
 * recreated, MGX_DIST_SPEC is set in its environment only, a traversal threw there) makes every rank go level by level instead of
 * issuing different RCCL collectives.  mgx_dbfs2_forget_plan drops this engine's history (what a recreated handle starts with). */
MGX_API int mgx_dbfs2_forget_plan(mgx_dbfs2_t h);
/* The engines of ALL ranks of one partition, made on one context, driven in turn by the calling thread: the collectives are device
 * copies into one shared buffer, the level plan is mgx_dbfs2_run's.  A measurement and test entry (wall time / ranks = what one
 * rank's GPU spends per traversal, no exchange time); out6_each: ranks x 6, as mgx_dbfs2_status per engine. */
MGX_API int mgx_dbfs2_run_group(mgx_dbfs2_t* engines, int count, int src_global, int64_t exchange_words, int64_t* out6_each);

/* ---- SSSP: sssp_problem_t / sssp_functor_t / sssp_enactor_t (gunrock/src/sssp/) ---- */
MGX_API int mgx_sssp_create(mgx_graph_t g, int src, mgx_sssp_t* out);     /* sssp_problem.hxx:40-52 */
MGX_API int mgx_sssp_reset(mgx_sssp_t p, int src);
MGX_API int mgx_sssp_free(mgx_sssp_t p);
MGX_API int mgx_sssp_distances(mgx_sssp_t p, float* host_dist);           /* extract() :54-57       */
/* After mgx_sssp_enact: the functor's preds (sssp_functor.hxx:31-34; racy as the reference's).  After mgx_sssp_run (the fused loop
 * keeps none): a shortest-path TREE built from the final distances the first time it is asked for (include/mgx/sssp_preds.hpp) --
 * pred[v] = the largest u with a tight edge u -> v from a strictly nearer vertex; equal-distance ties (zero weights) are resolved in
 * rounds so that no cycle can form; pred[source] = pred[unreached] = -1.  mgx_sssp_build_preds builds them without copying:
 * stats2 = { equal-distance tight edges found, rounds }. */
MGX_API int mgx_sssp_preds(mgx_sssp_t p, int* host_preds);
MGX_API int mgx_sssp_build_preds(mgx_sssp_t p, int64_t* stats2);
MGX_API int mgx_sssp_distances_device(mgx_sssp_t p, float** d_dist);
/* advance_forward_kernel<sssp_problem_t,sssp_functor_t,false,true> (sssp_enactor.hxx:49-54) */
MGX_API int mgx_sssp_advance(mgx_sssp_t p, mgx_frontier_t in, mgx_frontier_t out, int iteration, int64_t* front);
/* filter_kernel<sssp_problem_t,sssp_functor_t>                     (sssp_enactor.hxx:60-65) */
MGX_API int mgx_sssp_filter(mgx_sssp_t p, mgx_frontier_t in, mgx_frontier_t out, int iteration, int64_t* kept);
/* sssp_enactor_t::enact (sssp_enactor.hxx:40-72).  stats[0]=iterations, [1]=edge relaxations,
 * [2]=sum of input frontier lengths.                                                      */
MGX_API int mgx_sssp_enact(mgx_sssp_t p, float queue_sizing, int64_t* stats);
/* fused device-resident SSSP (same fixed point): stats as above                          */
MGX_API int mgx_sssp_run(mgx_sssp_t p, int src, int64_t* stats);
/* the same with near / far buckets of width delta (the delta-stepping BASELINE config 3 names): improved vertices whose
 * distance lies at or above the current threshold wait until the queue of nearer ones has run dry, then the threshold
 * moves to the bucket of the smallest waiting distance.  Same fixed point, fewer relaxations, more (cheap) iterations.
 * delta == 0: off (plain frontier Bellman-Ford, what mgx_sssp_run does); delta < 0: the default (off).  stats as above ([0] counts
 * the iterations incl. the bucket changes). */
MGX_API int mgx_sssp_run_delta(mgx_sssp_t p, int src, float delta, int64_t* stats);
/* per-launch timing of the relax kernel of mgx_sssp_run (k_sssp_relax: sssp_functor.hxx:20-29 over every edge of the
 * frontier): on != 0 brackets each launch with HIP events on the context's stream (each costs ~6 us of stream gap: measurement
 * runs only); mgx_sssp_kernel_times: out2 = { launches, device ns } of the last run                                        */
MGX_API int mgx_sssp_set_kernel_timing(mgx_sssp_t p, int on);
/* per-iteration trace of the last mgx_sssp_run (first min(cap, 63) iterations): frontier vertices, edges relaxed, duration in
 * ms from device-side timestamps taken when an iteration is opened (0 for the last one); *iterations = how many there were */
MGX_API int mgx_sssp_iteration_trace(mgx_sssp_t p, int cap, int64_t* frontier, int64_t* edges, float* ms, int* iterations);
MGX_API int mgx_sssp_kernel_times(mgx_sssp_t p, int64_t* out2);

/* ---- PR: pr_problem_t / pr_functor_t / pr_enactor_t (gunrock/src/pr/) ---- */
MGX_API int mgx_pr_create(mgx_graph_t g, int max_iter, mgx_pr_t* out);    /* pr_problem.hxx:33-44   */
MGX_API int mgx_pr_free(mgx_pr_t p);
MGX_API int mgx_pr_enact(mgx_pr_t p, int64_t* frontier_len_per_iter, int* iterations); /* pr_enactor.hxx:41-79 */
MGX_API int mgx_pr_ranks(mgx_pr_t p, float* host_ranks);

/* ---- k-core: kcore_problem_t / kcore_functor.hxx / kcore_enactor_t (gunrock/src/kcore/) ---- */
MGX_API int mgx_kcore_create(mgx_graph_t g, mgx_kcore_t* out);             /* kcore_problem.hxx:37-49 */
MGX_API int mgx_kcore_reset(mgx_kcore_t p);                                /* core numbers 0, degrees = row lengths */
MGX_API int mgx_kcore_free(mgx_kcore_t p);
/* kcore_enactor_t::enact (kcore_enactor.hxx:40-86): peeling on filter<deg_less_than_k> / advance<update_deg, false, false> /
 * filter<deg_atleast_k>.  *largest_k_core: what the enactor found (-1 if no k <= n ended the run: only a graph without
 * entries, an upstream quirk that is kept).  stats[0] = k values tried, [1] = passes, [2] = entries expanded,
 * [3] = vertices removed (stats may be NULL).  The run consumes the working degrees: mgx_kcore_reset before another one. */
MGX_API int mgx_kcore_enact(mgx_kcore_t p, int* largest_k_core, int64_t* stats);
MGX_API int mgx_kcore_num_cores(mgx_kcore_t p, int* host_num_cores);       /* extract() :51-53        */
MGX_API int mgx_kcore_degrees(mgx_kcore_t p, int* host_degrees);           /* the working degrees (<= 0 once a run is over) */

/* ---- synthetic input: counter-based R-MAT (the reference ships none, F4) ----
 * Writes edges [first_edge, first_edge+count) of the (scale, seed) stream to device arrays. */
MGX_API int mgx_rmat_edges(mgx_ctx_t ctx, int scale, int64_t first_edge, int64_t count, uint64_t seed,
                           int scramble, int* d_src, int* d_dst, float* d_weight);

#ifdef __cplusplus
}
#endif
#endif /* MGX_H_ */
