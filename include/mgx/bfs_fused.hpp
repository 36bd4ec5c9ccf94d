// mgx/bfs_fused.hpp -- device-resident BFS: advance + filter fused per level, no host round trip per
// level.  This header holds the shared pieces: control block, level bookkeeping, the deferred-mark epilogue, the
// kernels that build the next level's queues, the per-BFS state.  The bodies that expand a level live in
//   bfs_fused_stream.hpp  long rows, queue walk            bfs_fused_dense.hpp   long rows from the unit blocks
//   bfs_fused_wave.hpp    short rows, search per edge rank  bfs_fused_vshort.hpp  short rows vertex by vertex
//   bfs_fused_cold.hpp    the long rows' entries behind the LDS prefix, as pairs by slice of the id range
//   bfs_fused_chain.hpp   small levels, one workgroup       bfs_fused_pull.hpp    bottom-up levels
// and bfs_fused_run.hpp launches them, slot by slot (k_bfs_push: all bodies in ONE grid; k_bfs_chain_inplace).
//
// What the reference does per level: degree scan (K1) -> 4-byte D2H (K2) ->
// load-balanced expand writing one int per EDGE, mostly -1 (K3) -> compaction upsweep + D2H (K4)
// -> downsweep (K5); ~10 launches, 2 host syncs, 4-5 cudaMalloc/cudaFree pairs, and
// 16 B/edge + 40 B/vertex of traffic.  Here a level is
//
//   bfs_open_level      bookkeeping (sizes, termination flag, TEPS numerator, direction): one thread -- of the slot's
//                       push grid, of a chain of small levels, or of k_bfs_level_begin (partitioned runs).
//   push bodies         MARK ONLY: a neighbour that is not in the visited bitmap gets mark[v] = 1, a plain byte
//                       store.  No atomics on the hot path: measured on MI355X, device-scope atomics execute at the
//                       memory side (the per-XCD L2s are not coherent with each other), drop their L2 line,
//                       and together with the re-reads of the lines they dropped ran at ~5 G/s in a level with
//                       1.7 M claims -- 0.34 ms of a 0.40 ms kernel.  Byte stores are idempotent (every writer
//                       writes the same value), merge in the write-back L2s by byte mask, and are visible to
//                       the next kernel; nobody reads them while the level runs.  The visited bitmap itself is
//                       read-only during a level, so its hot prefix can sit in LDS and the rest in L2.  (Marks are
//                       not free either -- a scattered store costs the fabric a cache line, tools/microbench4.hip:
//                       hence the deferred hot marks below and the cold-edge pass.)
//   k_bfs_build2        one sweep over mark[] and the bitmap (+ the bitmaps the push workgroups flushed): marked and
//                       not yet visited = the level's discoveries.  Sets their bits (a thread owns the half-word of
//                       its 16 vertices: plain stores), writes their labels, the frontier bitmap and -- unless the
//                       next slot will not read them (bfs_build_is_lazy) -- the next level's two queues: one packed
//                       scan per queue, ONE cursor atomic per workgroup and queue (a hot 64-bit cursor takes only
//                       ~83 M returning atomics/s).  k_bfs_build is the list-based variant (partitioned runs).
//
// Queues: a frontier is stored as (row_start, scanned_edge_offset) pairs: a batch is appended with ONE 64-bit
// atomicAdd on a packed (vertex_count << 38 | edge_count) cursor, so the slot it gets back is at once the
// queue position and the exclusive degree scan of that position -- the scan the reference recomputes every
// level (K1) comes for free, and the next level can cut its edges into equal slices.  There are two queues per
// level: rows of at least args.long_min edges (streamed row-wise; their offsets count the degrees rounded up to
// 64, bfs_lq_*) and the rest (searched per edge rank).  Zero-degree discoveries are labelled but never queued.
//
// Algorithmic traffic: 8 B per traversed edge (col index + visited/label probe) and 20 B per
// frontier vertex -- the figure BASELINE.md's roofline uses.
#pragma once
#include <cstdlib>
#include <cstring>

#include "runtime.hpp"
#include "env.hpp"
#include "wave.hpp"

namespace mgx {

constexpr int BFS_MAX_TRACE = 4096;            // per-level trace slots
constexpr int BFS_VSHIFT = 38;                 // cursor = (vertices << 38) | edges
constexpr u64 BFS_EMASK = (1ull << BFS_VSHIFT) - 1ull;

struct bfs_ctrl_t {
  // The four words k_bfs_build's workgroups add to, each on a 128-byte line of its own: device-scope atomics on one
  // line are served one after the other (~12 ns each, profiles/r01/microbench.jsonl: hot_counter), and 500 workgroups
  // arrive at them at about the same time.
  u64 cursor[3];     // short-row queue: level L reads [L%3], level L's build fills [(L+1)%3], begin clears [(L+2)%3]
  u64 pad_a_[13];
  u64 lcursor[3];    // long-row queue (rows of degree >= args.long_min), same packing and rotation -- but its
                     // "edges" are the degrees rounded up to 64 (see bfs_lq_* below) ...
  u64 pad_b_[13];
  u64 ledges[3];     // ... and these are the true ones
  u64 pad_c_[13];
  u64 reached;       // vertices labelled (incl. source and zero-degree discoveries)
  u64 pad_d_[15];
  u64 merged_new;    // partitioned BFS (bfs_dist2.hpp): vertices discovered by ALL ranks in the level just merged
  u64 sum_edges;     // sum over levels of E  == m_t (out-degrees of reached vertices)
  u64 sum_frontier;  // sum over levels of frontier sizes (reached vertices with degree >= 1)
  u64 sum_long_edges;     // the part of sum_edges / sum_frontier that went through the long-row queue
  u64 sum_long_vertices;
  u64 claims;        // mark stores issued (>= reached-1: several edges may mark the same vertex)
  u64 claims_level[64];
  u64 pull_edges;    // in-edges inspected by bottom-up levels
  int done;
  int levels;        // number of levels that expanded at least one edge
  int pull;          // direction of the level about to run (set by k_bfs_level_begin; sticky once 1)
  int push_levels;   // levels run top-down
  int d2_append_level;     // partitioned runs: the level whose push appended its discoveries to the rank's id list itself (bfs_fused_sparse.hpp: k_d2_newbits has nothing to sweep); -1: none
  int d2_declared_level;   // partitioned runs: the level whose sweep declared its list overflowed without filling it (k_d2_newbits); -1: none
  int small_levels;  // levels run by the chains of small levels (bfs_fused_chain.hpp)
  int slots;         // launch slots that found work (their opener / chain counts them)
  int dist_done;     // partitioned runs: a level ended with no discovery on any rank ...
  int dist_levels;   // ... and this many levels hold vertices
  // Slot scheme (bfs_fused_run.hpp): launch slot s = [push, (pull), build]; the rings above are indexed by SLOT there
  // (s % 3), the queue buffers by s & 1, and the level a slot works on is slot_level[s & 3] -- written one slot ahead
  // (by the slot's opener, or by the chain of small levels that ran inside the push launch), so that nothing a
  // workgroup bases its decisions on changes while the slot's kernels run.
  int slot_level[4];
  int skip_build[4]; // slot s & 3: the push launch ran its level(s) itself (bfs_fused_chain.hpp): k_bfs_build has nothing to do
  // Deferred hot marks (bfs_hot_epilogue below): flush_count[s & 1] = workgroups of slot s that wrote their discoveries
  // as a bitmap into flush buffer 0 .. count - 1 instead of storing marks; zeroed one slot ahead by the opener.
  u32 flush_count[2];
  int fb_slot;       // frontier_bits holds exactly the frontier of this slot (written by the k_bfs_build of the slot before)
  // fused SSSP with near / far buckets (sssp_fused.hpp): the current threshold (float bits) and, per iteration parity,
  // how many improved vertices the queue build left for a later bucket and the smallest of their distances
  u32 sssp_thr;
  u32 sssp_far_cnt[2];
  u32 sssp_far_min[2];
  int dense_slots;   // slots whose long rows were read from the unit blocks (bfs_fused_dense.hpp)
  int vshort_slots;  // slots whose short rows were walked vertex by vertex (bfs_fused_vshort.hpp)
  int lazy_slot;     // the slot whose queues were NOT written (bfs_build_is_lazy): it must take both queue-less bodies; -1: none
  int lazy_slots;    // how many there were
  int cold_slot;     // the slot whose push ran the cold-edge pass (bfs_fused_cold.hpp): its build ORs the cold bitmaps; -1: none
  int cold_slots;    // how many there were
  int mini_slots;    // levels expanded by M launches (bfs_fused_mini.hpp)
  u32 mini_blocks[4];   // workgroups of the M launch of slot s & 3 that are through (a forwarding launch's last one moves the flags)
  u64 reached_mini;  // vertices labelled by M launches: NOT in `reached` (which must not change while such a launch decides)
  // Partitioned runs with a SPECULATIVE level plan (bfs_dist2.hpp: d2_run enqueues a whole traversal, lists or bitmaps per level
  // as the last traversals went, and looks once): a level that was planned for id lists but overflowed one FREEZES the
  // traversal -- every kernel of a later level returns at once (bfs_d2_frozen) -- until the host has sent the level through the
  // bitmap exchange.  d2_level_kind[L]: 1 merged from id lists, 2 its lists overflowed, 0 bitmaps without asking (or not run);
  // d2_level_new[L]: vertices all ranks discovered in level L (what the next plan is made from).
  int d2_frozen_level;           // -1: not frozen
  unsigned char d2_level_kind[64];
  u32 d2_level_new[64];
  u64 stamp[64];     // s_memrealtime (100 MHz) when each level was opened: per-level times without host syncs
  u64 trace[BFS_MAX_TRACE];   // (vertices << 38 | edges) of each level, both queues (kept LAST: read back up to `levels`)
};

constexpr int BFS_COLD_MAX_SLICES = 64;           // slices of the id range that may hold cold-edge pairs (bfs_fused_cold.hpp; 64: a rank of RMAT-26 / 8 uses 51)

struct bfs_fused_args_t {
  const u32* row_offsets;
  const int* col_indices;
  int* labels;
  u32* visited;        // (n+31)/32 words; written only by k_bfs_build, between levels: read-only while a level runs
  unsigned char* mark; // n bytes: set by the traversal kernels for every neighbour that may be new
  u32* frontier_bits;  // direction-optimising runs: bitmap of the level's frontier (k_bfs_build: the discoveries)
  u32* fr_row[2];      // short-row queue: CSR row start of each frontier vertex
  u32* fr_off[2];      //                  exclusive scan of degrees (edge rank of its first edge)
  u32* lq_row[2];      // long-row queue, same layout
  u32* lq_off[2];
  bfs_ctrl_t* ctrl;
  const u32* in_offsets;   // in-edges for bottom-up levels (== row_offsets/col_indices on symmetric graphs)
  const int* in_indices;
  const int* old_of_new;   // hub-first layout: original id of layout vertex v (NULL = identity)
  const int* new_of_old;   // and its inverse
  int long_min;            // rows of at least this many edges go to the long-row queue (0: no such queue)
  u32 hot_min_edges;       // a push kernel copies the hot prefix of the bitmap into LDS when its queue holds at least this many edges
  int mode;                // MGX_BFS_PUSH / MGX_BFS_DIRECTION_OPT
  float alpha;             // switch to bottom-up when unvisited < frontier_vertices * alpha (bfs_enactor.hxx:68)
  int n;
  int count_marks;         // the push kernels count their mark stores into ctrl->claims / claims_level (tools only)
  // unit blocks of the long rows (mgx_layout.hip: rows of >= long_min edges padded to 64-entry units; NULL: none)
  const int* ub_col;       // units_pad * 64 entries + 4 x (-1)
  const u32* ub_col24;     // the same, 24 bits per entry (12 bytes per four entries); NULL: not available
  const int* ub_owner;     // units_pad owners (vertex id in the space of row_offsets; n for padding units)
  u32 ub_units;            // real units
  u32 ub_units_pad;        // multiple of 16
  u32 ub_hot_only;         // ub_col24 / ub_owner / ub_units_pad are the blocks WITHOUT the entries of the cold-edge lists: for the slots that run
                           // the cold-edge pass.  A slot that reads unit blocks without the pass (a sparse level behind a lazy build) takes the
                           // full blocks: ub_col (32 bits) or ubf_col24, ubf_owner, ubf_units_pad (bfs_dense_body)
  const u32* ubf_col24;
  const int* ubf_owner;
  u32 ubf_units_pad;
  u32 dense_div;           // a slot reads its long rows from the unit blocks when frontier units * dense_div >= ub_units (0: never)
  // short rows vertex by vertex (bfs_fused_vshort.hpp; needs a degree-sorted CSR with 8 readable ints behind col_indices)
  u32 vs_v[4];             // class boundaries: [0]..[1] degrees 17..long_min-1, [1]..[2] 5..16, [2]..[3] 1..4
  u32 vs_v9;               // inside [1]..[2]: where the degrees drop below 9 (== vs_v[2]: not known, 5..16 is one class of four lanes per vertex)
  u32 vs_edges;            // edges of the rows in [vs_v[0], vs_v[3])
  u32 vs_div;              // a slot takes its short rows this way when its short-row queue holds >= vs_edges / vs_div edges (0: never)
  u32 vs_dummy;            // index (into col_indices) of four entries of -1
  u32* flush_buf;          // BFS_FLUSH_MAX buffers of BFS_FLUSH_WORDS words (NULL: hot marks are never deferred)
  u32 defer_min_marks;     // a workgroup with more deferred discoveries than this flushes them as a bitmap (else: byte marks)
  u32 defer_words;         // words of the LDS prefix whose marks are deferred (a multiple of 32, <= BFS_FLUSH_WORDS: the flush buffers' stride)
  u32 defer_reach_mul, defer_reach_div;   // a level defers while reached * mul < deferred range * div (1 / 1; MGX_BFS_DEFER_REACH="mul/div")
  int lazy_pull;           // direction-optimising runs: the builds behind bottom-up levels write no queues
  int merged_pull;         // direction-optimising runs: the bottom-up sweep runs inside the push launch (no k_bfs_pull_level launch)
  u32 chain_max_edges;     // a level of at most this many edges (and BFS_CHAIN_CQ rows) runs inside block 0 of the push launch (0: never)
  u32 chain_big_edges;     // ... and of at most this many edges inside the in-place chain kernel (k_bfs_chain_inplace; 0: never)
  u32 lazy_div;            // the build behind a level with >= n / lazy_div mark stores writes no queues (bfs_build_is_lazy; 0: never)
  u32* slot_marks;         // [2][BFS_MARK_CTRS] counters, 128 bytes apart: marks stored by the push workgroups of slot s in set s & 1 (NULL: not counted)
  // cold-edge lists of the long rows (bfs_fused_cold.hpp, mgx_layout.hip): (owner, dst) pairs of the unit blocks' entries
  // that point behind the LDS prefix, grouped by slice of the id range; NULL: none (the unit-block body marks them)
  const int* cold_owner;
  const int* cold_dst;
  int cold_slices;         // slices that hold pairs (<= BFS_COLD_MAX_SLICES)
  u32 cold_lo[BFS_COLD_MAX_SLICES];         // first vertex of slice i (a multiple of 1024; the slice is BFS_COLD_WORDS * 32 vertices)
  u32 cold_off[BFS_COLD_MAX_SLICES + 1];        // its pairs: [cold_off[i], cold_off[i + 1])
  u32 cold_wgs[BFS_COLD_MAX_SLICES + 1];        // the cold workgroups [cold_wgs[i], cold_wgs[i + 1]) of a push launch take slice i
  u32* cold_flush;         // cold_wgs[cold_slices] bitmaps of BFS_COLD_WORDS words: what cold workgroup k discovered in its slice
  // (round 6) a FLAT graph: the lists hold EVERY entry of every row (slices from vertex 0 on).  A level that holds at least an eighth
  // of them (cold_all_pairs) is one sweep of the lists by the cold workgroups -- nothing else of the push grid works --, the other
  // levels walk their queues and mark what they find untested.  0: the lists are the long rows' entries behind the LDS prefix
  u32 cold_all = 0;
  u64 cold_all_pairs = 0;
  // the same pairs at four bytes each (mgx_layout.hip: mgx_cold_pack_device): bits 0..19 dst - cold_lo[slice], bits 20..31
  // (owner - the owner of the first pair of the pair's 64-chunk) / cold_ranks; NULL: none.  Slice q is packed when bit q of
  // cold_pk_mask is set; its chunks' owners are cold_cbase[cold_cb[q] ..)
  const u32* cold_pk = nullptr;
  const u32* cold_cbase = nullptr;
  u64 cold_pk_mask = 0;
  u32 cold_ranks = 1;
  u32 cold_cb[BFS_COLD_MAX_SLICES + 1] = {};
  // a rank of the partitioned traversal (bfs_dist2.hpp; all NULL / 0 on the single-GPU path): its id list of the level
  // ([0] count, ids from D2_LIST_HEAD on), the list's capacity in ids, the rank's new-bit map (bfs_fused_sparse.hpp)
  u32* d2_list = nullptr;
  u32 d2_list_cap = 0;
  u32* d2_newbits = nullptr;
  u32* d2_front = nullptr;         // the level's frontier over the rank's LOCAL rows (k_bfs_build2 writes it, the vertex-by-vertex body reads it)
  const int* vs_col = nullptr;     // the vertex-by-vertex body's copy of col_indices with readable entries behind it (NULL: col_indices has them)
  // cold-edge lists of the SHORT rows (the entries the vertex-by-vertex body would mark untested), the long rows' slices: the same pass
  // takes them on a level that walks its short rows vertex by vertex.  Equal to marking on RMAT-22 (6 % of the entries), what the big
  // graphs need: RMAT-25 2.31 -> 2.10 ms per traversal, RMAT-24 1.153 -> 1.127 (round 5; a lab shape until then).  NULL: none
  const int* colds_owner = nullptr;
  const int* colds_dst = nullptr;
  u32 colds_off[BFS_COLD_MAX_SLICES + 1] = {};
#ifdef MGX_LAB
  // ---- lab build only (-DMGX_LAB, never set by __graft_entry__.build()): shapes that lost their A/B runs and the
  // instrumented kernels of the measurement tools.  The product library carries none of this: MGX_LAB_GET reads a
  // constant there and the compiler drops the code behind it.
  int flags;               // MGX_BFS_FLAGS: instrumented stream kernel (results are wrong by design)
  int build_diag;          // measurements only (MGX_BFS_BUILD_DIAG; results wrong): 1 no label stores, 2 no extent gathers, 4 no cursor atomics, 8 no queue stores, 16 synthetic extents, 32 / 64 no OR of the deferred / the cold pass's bitmaps
  int dense_diag;          // measurements only (MGX_BFS_DENSE_DIAG): 1 the unit-block body stores no marks, 2 tests nothing
#endif
};

// a lab-only field of the arguments / run options: the field in a lab build, a constant in the product
#ifdef MGX_LAB
#define MGX_LAB_GET(obj, field, product_value) ((obj).field)
#else
#define MGX_LAB_GET(obj, field, product_value) (product_value)
#endif

// Copy `words` words (a multiple of 4; src 16-byte aligned, readable up to the next multiple of 4 * NT words... clamped)
// of the bitmap into LDS: ALL of a thread's 16-byte loads are issued before the first one is stored.  (The plain loop
// `for (i = tid; i < q; i += NT) dst[i] = src[i]` compiles to load - wait - store per trip: five dependent round trips
// to L2 at the start of every workgroup of every level.)
template <int NT, int WORDS>
__device__ __forceinline__ void bfs_copy_prefix(u32* __restrict__ dst, const u32* __restrict__ src) {
  static_assert(WORDS % 4 == 0, "whole 16-byte pieces");
  constexpr int Q = WORDS / 4;
  constexpr int IT = (Q + NT - 1) / NT;
  const uint4* const s4 = (const uint4*)src;
  uint4* const d4 = (uint4*)dst;
  uint4 v[IT];
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const int i = k * NT + (int)threadIdx.x;
    v[k] = s4[i < Q ? i : Q - 1];
  }
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const int i = k * NT + (int)threadIdx.x;
    if (i < Q) d4[i] = v[k];
  }
}

constexpr int BFS_COLD_WGS = 64;               // workgroups of a push launch that take the cold pairs: at least this many ...
constexpr int BFS_COLD_WGS_MAX = 1024;         // ... one per 131 072 pairs, at most this many (cold_wgs[cold_slices] says how many).  Every one of them costs a copy of its slice into LDS, an 80 KB bitmap to write and to OR (RMAT-22, 9.4 M pairs: 143 workgroups 0.3017 ms per traversal, 64-96 0.2983-0.3006, 192 0.3055)
constexpr int BFS_COLD_WORDS = 20384;          // bitmap words of a slice == the unit-block body's LDS prefix (BFS_DENSE_HOTW)

constexpr int BFS_MARK_CTRS = 8;               // the workgroups of a push launch spread their adds over this many lines
constexpr int BFS_MARK_STRIDE = 32;            // u32 words between two counters

// End of a push body (all threads of the workgroup): its mark count goes to the slot's counters -- what the queue build
// bases its lazy decision on -- and, for the tools, to the statistics.  s_int[0] must be 0 on entry.
__device__ __forceinline__ void bfs_body_finish(const bfs_fused_args_t& a, int marks, int slot, int stat_level, int* s_int) {
  if (!a.slot_marks && !a.count_marks) return;
  marks = wave_sum(marks);
  if ((threadIdx.x & (WAVE - 1)) == 0 && marks) atomicAdd(&s_int[0], marks);
  __syncthreads();
  if (threadIdx.x == 0 && s_int[0]) {
    if (a.slot_marks) atomicAdd(&a.slot_marks[((slot & 1) * BFS_MARK_CTRS + (int)(blockIdx.x % BFS_MARK_CTRS)) * BFS_MARK_STRIDE], (u32)s_int[0]);
    if (a.count_marks) {           // statistics for the tools: two device-scope atomics per workgroup on one line
      atomicAdd(&a.ctrl->claims, (u64)s_int[0]);
      if (stat_level < 64) atomicAdd(&a.ctrl->claims_level[stat_level], (u64)s_int[0]);
    }
  }
}
__device__ __forceinline__ void bfs_slot_marks_clear(const bfs_fused_args_t& a, int slot) {       // one thread
  if (a.slot_marks)
    for (int i = 0; i < BFS_MARK_CTRS; ++i) a.slot_marks[((slot & 1) * BFS_MARK_CTRS + i) * BFS_MARK_STRIDE] = 0u;
}

__device__ __forceinline__ void bfs_ctrl_reset(bfs_ctrl_t* c) {
  for (int i = 0; i < 3; ++i) { c->cursor[i] = 0; c->lcursor[i] = 0; c->ledges[i] = 0; }
  c->merged_new = 0;
  c->sum_edges = c->sum_frontier = c->sum_long_edges = c->sum_long_vertices = 0;
  c->reached = 1;
  c->claims = 0;
  for (int i = 0; i < 64; ++i) c->claims_level[i] = 0;
  c->pull_edges = 0;
  c->done = c->levels = c->pull = c->push_levels = 0;
  c->d2_append_level = c->d2_declared_level = -1;
  c->small_levels = c->slots = 0;
  c->dist_done = c->dist_levels = 0;
  for (int i = 0; i < 4; ++i) { c->slot_level[i] = 0; c->skip_build[i] = 0; }
  c->flush_count[0] = c->flush_count[1] = 0;
  c->fb_slot = 0;                                  // k_bfs_fused_init seeds frontier_bits with the source
  c->dense_slots = 0;
  c->vshort_slots = 0;
  c->lazy_slot = -1;
  c->lazy_slots = 0;
  c->cold_slot = -1;
  c->cold_slots = 0;
  c->mini_slots = 0;
  for (int i = 0; i < 4; ++i) c->mini_blocks[i] = 0u;
  c->reached_mini = 0;
  c->d2_frozen_level = -1;
  for (int i = 0; i < 16; ++i) ((u32*)c->d2_level_kind)[i] = 0u;
  for (int i = 0; i < 64; ++i) c->d2_level_new[i] = 0u;
  c->sssp_thr = 0x7f7fffffu;
  c->sssp_far_cnt[0] = c->sssp_far_cnt[1] = 0;
  c->sssp_far_min[0] = c->sssp_far_min[1] = 0x7f7fffffu;
  for (int i = 0; i < 64; ++i) c->stamp[i] = 0;
}

// Long-row queue entries.  The stream kernel (bfs_fused_stream.hpp) reads a row in sub-rounds of up to 64 consecutive
// edges, so a row of d edges costs ceil(d / 64) sub-rounds whatever d is: a slice of 4096 edges made of 65-edge rows is
// twice the work of one inside a hub row (measured on RMAT-22's big level: 44 rounds for the slowest wave against
// 26.7 on average, and the slowest wave is the kernel's duration).  The queue's offsets are therefore prefix sums of
// the PADDED degrees (multiples of 64: one unit = one sub-round) and equal slices are equal work; the low 6 bits of
// an entry, free in a multiple of 64, carry degree & 63 so that the true row end can be recovered:
//   entry i = P_i | (d_i & 63),  P_{i+1} - P_i = pad64(d_i),  d_i = P_{i+1} - P_i - ((64 - (entry_i & 63)) & 63)
__host__ __device__ __forceinline__ u32 bfs_lq_pad(u32 deg) { return (deg + 63u) & ~63u; }
__host__ __device__ __forceinline__ u32 bfs_lq_degree(u32 entry, u32 next_entry) {
  return ((next_entry & ~63u) - (entry & ~63u)) - ((64u - (entry & 63u)) & 63u);
}

// level-0 queue entry of the source (its row is `row`, e.g. a local row of a partition)
__device__ __forceinline__ void bfs_seed_queue(const bfs_fused_args_t& a, u32 row) {
  bfs_ctrl_t* c = a.ctrl;
  const u32 ro = a.row_offsets[row];
  const u32 deg = a.row_offsets[row + 1] - ro;
  const bool is_long = a.long_min > 0 && deg >= (u32)a.long_min;
  (is_long ? a.lq_row : a.fr_row)[0][0] = ro;
  (is_long ? a.lq_off : a.fr_off)[0][0] = is_long ? (deg & 63u) : 0u;
  (is_long ? c->lcursor : c->cursor)[0] = deg ? ((1ull << BFS_VSHIFT) | (u64)(is_long ? bfs_lq_pad(deg) : deg)) : 0ull;
  if (is_long) c->ledges[0] = deg;
}

// Start of a traversal, one launch: clears labels (-1), bitmap(s) and marks, and seeds the source.  The thread that
// clears the element holding the source's label / bit writes the seed value instead, so there is no ordering
// between workgroups to worry about.
// prev_ctrl / prev_head (batches of sources, bfs_fused_run_many): the head of the control block as the PREVIOUS traversal of
// the batch left it goes to (pinned) host memory first -- by the workgroup whose thread 0 then resets it; nobody else
// touches the block here.
__global__ __launch_bounds__(BLOCK) void k_bfs_fused_init(bfs_fused_args_t a, int src, long long nwords, const bfs_ctrl_t* prev_ctrl,
                                                          bfs_ctrl_t* prev_head, int head_words) {
  if (prev_head && blockIdx.x == 0) {
    const u32* const from = (const u32*)prev_ctrl;
    u32* const to = (u32*)prev_head;
    for (int i = threadIdx.x; i < head_words; i += BLOCK) to[i] = from[i];
    __syncthreads();
  }
  const int src_old = src;
  if (a.new_of_old) src = a.new_of_old[src];           // labels live in ORIGINAL id space, everything else in layout space
  const long long tid = (long long)blockIdx.x * BLOCK + threadIdx.x;
  const long long nth = (long long)gridDim.x * BLOCK;
  const long long n = a.n;
  for (long long i = tid; i < (n + 3) / 4; i += nth) {              // 4 labels / 4 marks per step
    if (i * 4 + 4 <= n) {
      const long long j = i * 4;                                     // (no dynamic indexing of the vector: scratch)
      const int4 l = make_int4(src_old == j ? 0 : -1, src_old == j + 1 ? 0 : -1, src_old == j + 2 ? 0 : -1,
                               src_old == j + 3 ? 0 : -1);
      *(int4*)(a.labels + j) = l;
      *(u32*)(a.mark + i * 4) = 0u;
    } else {
      for (long long j = i * 4; j < n; ++j) { a.labels[j] = (j == src_old) ? 0 : -1; a.mark[j] = 0; }
    }
  }
  const u32 src_bit = 1u << (src & 31);
  for (long long w = tid; w < nwords; w += nth) {
    const u32 seed = (w == (src >> 5)) ? src_bit : 0u;
    a.visited[w] = seed;
    a.frontier_bits[w] = seed;
  }
  if (tid == 0) {
    bfs_ctrl_reset(a.ctrl);
    bfs_slot_marks_clear(a, 0);
    bfs_slot_marks_clear(a, 1);
    bfs_seed_queue(a, (u32)src);
  }
}

// Opening a level (one thread): termination flag, trace, TEPS numerator, direction decision.  Returns false when
// the frontier is empty (the traversal is over).
// `slot` indexes the rings (== level in the explicit-level scheme of the partitioned path).
__device__ __forceinline__ bool bfs_open_level(const bfs_fused_args_t& a, int level, int slot) {
  bfs_ctrl_t* const c = a.ctrl;
  const u64 cur = c->cursor[slot % 3];
  const u64 lcur = c->lcursor[slot % 3];
  const long long nf = (long long)(cur >> BFS_VSHIFT) + (long long)(lcur >> BFS_VSHIFT);
  const u64 long_edges = c->ledges[slot % 3];
  const u64 E = (cur & BFS_EMASK) + long_edges;
  c->cursor[(slot + 2) % 3] = 0;
  c->lcursor[(slot + 2) % 3] = 0;
  c->ledges[(slot + 2) % 3] = 0;
  if (level < 64) c->stamp[level] = __builtin_amdgcn_s_memrealtime();
  if (nf == 0) {
    if (!c->done) { c->done = 1; c->levels = level; }
    return false;
  }
  if (level < BFS_MAX_TRACE) c->trace[level] = ((u64)nf << BFS_VSHIFT) | E;
  c->sum_edges += E;
  c->sum_frontier += (u64)nf;
  c->sum_long_edges += long_edges;
  c->sum_long_vertices += lcur >> BFS_VSHIFT;
  if (a.mode == 1 && !c->pull) {
    const float unvisited = (float)((long long)a.n - (long long)c->reached);
    if (unvisited < (float)nf * a.alpha) c->pull = 1;      // bfs_enactor.hxx:68; never switches back (:74-112)
  }
  if (!c->pull) c->push_levels += 1;
  return true;
}

// Is level `level` a bottom-up one?  The level's opener writes the decision to ctrl->pull, but in the direct launch
// scheme it runs inside the push grid, next to the workgroups that need the answer: they derive it themselves from
// the same, stable inputs (queue sizes left by the previous build, vertices reached so far) with the same arithmetic.
__device__ __forceinline__ bool bfs_level_pulls(const bfs_fused_args_t& a, const bfs_ctrl_t* c, int slot) {
  if (a.mode != 1) return false;
  if (c->pull) return true;
  const long long nf = (long long)(c->cursor[slot % 3] >> BFS_VSHIFT) + (long long)(c->lcursor[slot % 3] >> BFS_VSHIFT);
  const float unvisited = (float)((long long)a.n - (long long)c->reached);
  return unvisited < (float)nf * a.alpha;
}

// Explicit-level variant of the above as a kernel of its own (partitioned runs: the host counts the levels).
// There the traversal is over when the level before discovered nothing on ANY rank (ctrl->merged_new, the same
// number on every rank), whatever this rank's own queues hold.
__device__ __forceinline__ void bfs_begin_level(const bfs_fused_args_t& a, int level, int partitioned) {
  bfs_ctrl_t* const c = a.ctrl;
  if (partitioned && level > 0 && c->merged_new == 0 && !c->dist_done) { c->dist_done = 1; c->dist_levels = level; }
  (void)bfs_open_level(a, level, level);
}
__global__ void k_bfs_level_begin(bfs_fused_args_t a, int level, int partitioned) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  bfs_begin_level(a, level, partitioned);
}

// a rank's traversal is frozen below `level` (see bfs_ctrl_t::d2_frozen_level): grid-uniform, nothing of a later level may run
__device__ __forceinline__ bool bfs_d2_frozen(const bfs_ctrl_t* c, int level) {
  const int f = c->d2_frozen_level;
  return f >= 0 && level > f;
}

// Kernel argument `arg` of the level kernels: >= 0 is an explicit level (the partitioned path: the host counts the
// levels, slot == level); -1 - s is launch slot s of the slot scheme, whose level is ctrl->slot_level[s & 3].
__device__ __forceinline__ void bfs_resolve(const bfs_ctrl_t* c, int arg, int& slot, int& level) {
  if (arg >= 0) { slot = level = arg; return; }
  slot = -1 - arg;
  level = c->slot_level[slot & 3];
}
__host__ __device__ __forceinline__ int bfs_slot_arg(int slot) { return -1 - slot; }

// ---- deferred hot marks ------------------------------------------------------------------------------------------------
// A push workgroup claims the bit of a neighbour it finds unvisited in ITS LDS copy of the bitmap prefix (exact dedup
// inside the workgroup) -- but 512 workgroups each do, so a vertex that many edges of the level lead to is marked by most
// of them: a level of a few thousand hubs stored 11.3 M marks for 1.28 M discoveries (RMAT-22), and mark stores are
// bound by the L2s' request rate (~230 G/s: 50 us of that 110 us level; a discovery-heavy level spent 130 of its 210 us
// on them).  So the marks of the vertices in the first BFS_FLUSH_WORDS words of the prefix are DEFERRED: the claim in
// LDS is all a workgroup does while it runs; at its end it compares its LDS words with the bitmap it started from
// (read-only during the level) and either
//   * stores the few marks it has (no more than defer_min_marks: the same stores, later), or
//   * writes the difference as a bitmap, coalesced, into a flush buffer of its own (72 KB = 562 cache lines, instead of
//     tens of thousands of scattered one-byte requests), numbered by a ticket; k_bfs_build ORs the slot's flush buffers
//     into its sweep (all waves of a workgroup share the reads of a run: a few coalesced loads each).
// Vertices behind the deferred range, cold neighbours and small levels (no LDS copy) keep their immediate byte marks.
// The epilogue costs every workgroup a second read of its 72 KB of bitmap (~4 us per level, measured), which only pays
// while discoveries are dense: a level defers while fewer vertices have been reached than the deferred range holds
// (bfs_defer_limit: grid-uniform, ctrl->reached is stable while a level runs; a quarter of the range was the rule while
// the cold marks still dominated those levels) -- the hub levels at the start of a
// traversal; later levels find most of the prefix visited and store the few marks they have at once.
constexpr int BFS_FLUSH_WORDS = 20352;                 // 636 runs of 1024 vertices: (nearly) all of the unit-block body's LDS prefix; a body with a shorter prefix defers what it has
constexpr int BFS_FLUSH_MAX = 1024;                    // one buffer per push workgroup of a slot at most
constexpr int BFS_FLUSH_RUNS = BFS_FLUSH_WORDS / 32;

// vertices [0, limit) defer their marks in this level (0: nothing is deferred)
__device__ __forceinline__ u32 bfs_defer_limit(const bfs_fused_args_t& a, u32 hot_n) {
  if (!a.flush_buf || hot_n == 0u) return 0u;
  const u32 range = hot_n < a.defer_words * 32u ? hot_n : a.defer_words * 32u;
  return a.ctrl->reached * (u64)a.defer_reach_mul < (u64)range * (u64)a.defer_reach_div ? range : 0u;
}

// End of a push workgroup (all threads; contains barriers).  hot: the LDS copy, deferred words = min(hot words in use,
// BFS_FLUSH_WORDS); marks: this thread's count of claims (deferred and stored alike); s_red: 2 ints of LDS.
// A workgroup with many claims writes its LDS words AS THEY ARE -- the bitmap it started from plus its own claims; the
// queue build masks with the old bitmap anyway (new = (marks | flushed) & ~visited) -- so it never reads the global bitmap
// again (that second read was ~4 us of every deferring level).  Only a workgroup with few claims looks at the difference,
// to replay it as byte marks.
template <int NT>
__device__ __forceinline__ int bfs_hot_epilogue(const bfs_fused_args_t& a, const u32* hot, u32 defer_words, int slot,
                                                int* s_red, int marks) {
  if (defer_words == 0u) return 0;
  constexpr int PERT = (BFS_FLUSH_WORDS + NT - 1) / NT;          // words per thread (18 at 1024 threads)
  if (threadIdx.x == 0) { s_red[0] = 0; s_red[1] = -1; }
  __syncthreads();                                                // every wave's claims are in
  const int wm = wave_sum(marks);
  if (lane_id() == 0 && wm) atomicAdd(&s_red[0], wm);
  __syncthreads();
  const int total = s_red[0];                                     // claims of the workgroup (>= the deferred ones)
  if (total == 0) return 0;
  if (MGX_LAB_GET(a, dense_diag, 0) & 8) return total;            // (lab builds, measurements only: the deferred marks are dropped -- what a level costs without its flush)
  if ((u32)total > a.defer_min_marks) {
    if (threadIdx.x == 0) {
      const u32 k = atomicAdd(&a.ctrl->flush_count[slot & 1], 1u);
      s_red[1] = k < (u32)BFS_FLUSH_MAX ? (int)k : -1;            // (cannot overflow: one ticket per workgroup of the slot)
    }
    __syncthreads();
    const int k = s_red[1];
    if (k >= 0) {
      u32* const out = a.flush_buf + (size_t)k * BFS_FLUSH_WORDS;
#pragma unroll
      for (int q = 0; q < PERT; ++q) {
        const u32 i = (u32)q * NT + threadIdx.x;
        if (i < a.defer_words) out[i] = i < defer_words ? hot[i] : 0u;      // (the queue build reads the runs below the launch's deferred range: a body whose own range is shorter leaves zeros, never what an earlier level wrote)
      }
      return total;
    }
  }
  // few: the marks themselves.  All of this thread's words of the bitmap first, every load unconditional (round 5: with the load under
  // `i < defer_words` the code generator put each one in a branch of its own with an s_waitcnt vmcnt(0) behind it -- PERT dependent
  // round trips to the L2 at the end of the workgroup; a relaxed atomic load of wavefront scope is the same instruction, but stays put)
  constexpr int CH = 5;                                            // loads in flight per thread (all PERT at once cost the push kernels SGPR spills)
#pragma unroll
  for (int q0 = 0; q0 < PERT; q0 += CH) {
    u32 vis[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const u32 i = (u32)(q0 + k) * NT + threadIdx.x;
      vis[k] = __hip_atomic_load(a.visited + (i < defer_words ? i : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      if (q0 + k >= PERT) break;
      const u32 i = (u32)(q0 + k) * NT + threadIdx.x;
      u32 w = i < defer_words ? (hot[i] & ~vis[k]) : 0u;
      const u32 base = i * 32u;
      while (w) {
        const int b = __ffs((int)w) - 1;
        w &= w - 1u;
        a.mark[base + (u32)b] = 1;
      }
    }
  }
  return total;
}

// ---- a level's discoveries -> bitmap, labels, next level's queues ----------------------------------------------
// FROM_MARKS (single GPU): vertex v is new when mark[v] != 0 and its bit is not set in `visited`; the kernel sets
// the bit (and frontier_bits for direction-optimising runs).  Otherwise (partitioned runs): `bits` already holds
// the discoveries of all ranks, and this rank owns the vertices `local * ranks + rank`; rows and labels are
// addressed by the local index.  Every discovery gets label level+1; those with edges are appended to the
// short- or long-row queue of level+1.
// A workgroup owns BUILD_VPB vertices, 16 per thread: one 16-byte load of marks and one 16-bit load / store of
// the bitmap per thread (the half-word is the thread's own: plain stores), a workgroup scan of the counts,
// compaction into an LDS list, then per batch of up to BUILD_LIST discoveries: row extents, labels, and ONE
// 64-bit atomic per queue (slot and exclusive degree scan at once, see above).  The vertices of a workgroup are
// sixteen separate runs of 1024 (one per wave), a sixteenth of the id range apart: under the hub-first layout a
// level's discoveries are concentrated in a prefix of the ids, and contiguous ownership would leave most
// workgroups idle (measured: 64 us vs 26 us for two levels with the same number of discoveries).
struct __attribute__((aligned(4))) bfs_u32x2 { u32 x, y; };   // 8-byte load at 4-byte alignment

constexpr int BFS_BUILD_NT = 1024;
constexpr int BFS_BUILD_VPB = 16 * BFS_BUILD_NT;      // vertices per workgroup
constexpr int BFS_BUILD_LIST = 8 * BFS_BUILD_NT;      // discoveries appended per batch

inline int bfs_build_grid(long long n_local, int nt = BFS_BUILD_NT) { return (int)((n_local + 16 * nt - 1) / (16 * nt)); }

template <int NT, bool FROM_MARKS>
__global__ __launch_bounds__(NT, NT == 512 ? 4 : 4) void k_bfs_build(bfs_fused_args_t a, int arg, const u32* __restrict__ bits,
                                                  int* __restrict__ labels, int n_local, int ranks, int rank,
                                                  int stop_when_done) {
  constexpr int NW = NT / WAVE;
  constexpr int LIST = 8 * NT;
  constexpr int PER = LIST / NT;
  constexpr u64 CNT1 = 1ull << 40;
  constexpr u64 DEGMASK = CNT1 - 1ull;
  __shared__ u32 st_v[LIST];                     // the batch's discoveries, then their row starts
  __shared__ u32 st_d[LIST];                     // ... and degrees
  __shared__ u64 s_scan[NW + 1];
  __shared__ u64 s_base[2];
  __shared__ u32 s_long_edges;                   // true edges of the batch's long rows (their offsets count padded ones)
  bfs_ctrl_t* const c = a.ctrl;
  int slot, level;
  bfs_resolve(c, arg, slot, level);
  if (stop_when_done && c->done) return;        // (a rank of a partitioned run may be handed work with an empty frontier)
  if (!FROM_MARKS && bfs_d2_frozen(c, level)) return;
  if (arg < 0 && c->skip_build[slot & 3]) return;   // the push launch of this slot ran its level(s) itself
  if (FROM_MARKS && blockIdx.x == 0 && threadIdx.x == 0) c->fb_slot = slot + 1;   // frontier_bits: written in full below
  // first of this thread's 16 vertices: run (blockIdx + k * gridDim) of 1024 vertices, k = the thread's wave
  const long long i0 = (((long long)blockIdx.x + (long long)(threadIdx.x >> 6) * gridDim.x) * 64 + (threadIdx.x & 63)) * 16;
  const int new_label = level + 1;
  const int* __restrict__ old_of_new = a.old_of_new;
  const u32 long_min = a.long_min > 0 ? (u32)a.long_min : 0xFFFFFFFFu;

  // ---- deferred hot marks of the slot: the flush buffers' bits for this workgroup's runs (see bfs_hot_epilogue) -------
  u32 flushed16 = 0;
  if (FROM_MARKS && a.flush_buf) {
    const u32 F = (MGX_LAB_GET(a, build_diag, 0) & 32) ? 0u : c->flush_count[slot & 1];     // (lab builds, 32: the OR of the flushed bitmaps skipped)
    if (F) {                                                     // (grid-uniform)
      __shared__ u32 s_or[NW][WAVE];
      const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
      for (int hr = 0; hr < NW; ++hr) {
        const long long run = (long long)blockIdx.x + (long long)hr * gridDim.x;       // the run wave `hr` owns
        if (run >= (long long)(a.defer_words / 32u)) break;
        // all waves share the reads: wave w takes buffers w, w + NW, ...; a lane's halfword = its 16 vertices of the run
        const unsigned short* const col = (const unsigned short*)a.flush_buf + run * 64 + lane;
        u32 acc = 0;
#pragma unroll 16
        for (u32 k = (u32)wave; k < F; k += NW) acc |= col[(size_t)k * (BFS_FLUSH_WORDS * 2)];
        s_or[wave][lane] = acc;
        __syncthreads();
        if (wave == hr) {
#pragma unroll
          for (int w = 0; w < NW; ++w) flushed16 |= s_or[w][lane];
        }
        __syncthreads();
      }
    }
  }

  // ---- which of my 16 vertices are new ----------------------------------------------------------------------
  u32 new16 = 0;
  if (i0 < n_local) {
    const u32 valid = (n_local - i0 >= 16) ? 0xFFFFu : ((1u << (int)(n_local - i0)) - 1u);
    if (FROM_MARKS) {
      const uint4 m = *(const uint4*)(a.mark + i0);                          // mark[] is padded: always readable
      const u32 x[4] = {m.x, m.y, m.z, m.w};
      u32 m16 = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) m16 |= (((x[q] & 0x01010101u) * 0x10204080u) >> 28) << (4 * q);   // 4 marks -> 4 bits
      m16 |= flushed16;
      unsigned short* const vis16 = (unsigned short*)a.visited + (i0 >> 4);
      const u32 old16 = *vis16;
      new16 = m16 & ~old16 & valid;
      if (new16) *vis16 = (unsigned short)(old16 | new16);                   // this thread is the half-word's only writer
      ((unsigned short*)a.frontier_bits)[i0 >> 4] = (unsigned short)new16;     // bottom-up levels, unit blocks
    } else {
      for (int q = 0; q < 16; ++q) {
        const long long v = (i0 + q) * ranks + rank;
        if (((valid >> q) & 1u) && ((bits[v >> 5] >> (v & 31)) & 1u)) new16 |= 1u << q;
      }
    }
  }
  // ---- compaction: positions by a workgroup scan of the counts ------------------------------------------------
  u64 total64;
  const u64 before = block_exclusive_sum_lean<NW>((u64)__popc(new16), s_scan, &total64);
  const int total = (int)total64;
  if (total == 0) return;
  if (threadIdx.x == 0) atomicAdd(&c->reached, (u64)total);

  u64* const cur_s = &c->cursor[(slot + 1) % 3];
  u64* const cur_l = &c->lcursor[(slot + 1) % 3];
  u32* __restrict__ const out_row_s = a.fr_row[(slot + 1) & 1];
  u32* __restrict__ const out_off_s = a.fr_off[(slot + 1) & 1];
  u32* __restrict__ const out_row_l = a.lq_row[(slot + 1) & 1];
  u32* __restrict__ const out_off_l = a.lq_off[(slot + 1) & 1];

  for (int first = 0; first < total; first += LIST) {
    // my discoveries whose list position falls into [first, first + LIST)
    {
      u32 rest = new16;
      int at = (int)before - first;
      while (rest) {
        const int q = __ffs((int)rest) - 1;
        rest &= rest - 1;
        if (at >= 0 && at < LIST) st_v[at] = (u32)(i0 + q);
        ++at;
      }
    }
    if (threadIdx.x == 0) s_long_edges = 0;
    __syncthreads();
    const int cnt = (total - first < LIST) ? total - first : LIST;
    // Row extents, labels: STRIDED over the list (entry q * NT + thread), so that the lanes of a wave instruction
    // gather the extents and layout ids of 64 consecutive list entries -- neighbouring vertices, a few cache lines --
    // instead of 64 entries eight apart (the L2s are bound by requests, not bytes: this kernel's time is its gathers
    // and the label scatter).  The extents go back to LDS for the blocked pass below.
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = q * NT + threadIdx.x;
      if (i < cnt) {
        const u32 v = st_v[i];
        bfs_u32x2 ext;
        if (MGX_LAB_GET(a, build_diag, 0) & 2) { ext.x = v * 8u; ext.y = v * 8u + 3u; }
        else ext = *(const bfs_u32x2*)(a.row_offsets + v);                  // both ends of the row in one 8-byte load
        // (non-temporal stores here were measured slower.  So was moving this scatter out of the build -- vertex ids in
        //  the queues, labels written by 64 workgroups of the launch that consumes the queue: the builds of the two big
        //  RMAT-22 levels went from 30 to 23-25 us, but the push launches grew by more, 0.44 -> 0.46 ms per traversal:
        //  a million random 4-byte stores cost their ~20 us wherever they run, and here they have the most threads)
        if (!(MGX_LAB_GET(a, build_diag, 0) & 1)) labels[old_of_new ? old_of_new[v] : (int)v] = new_label;
        st_v[i] = ext.x;
        st_d[i] = ext.y - ext.x;
      }
    }
    __syncthreads();
    u32 ro[PER], dg[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = threadIdx.x * PER + q;
      ro[q] = (i < cnt) ? st_v[i] : 0u;
      dg[q] = (i < cnt) ? st_d[i] : 0u;
    }
    u64 loc[PER];
    u64 sum_s = 0, sum_l = 0;
    u32 longmask = 0, long_true = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const u32 deg = dg[q];
      const bool is_long = deg >= long_min;
      if (is_long) { longmask |= 1u << q; long_true += deg; }
      loc[q] = is_long ? sum_l : sum_s;
      const u64 add = deg ? (CNT1 | (u64)(is_long ? bfs_lq_pad(deg) : deg)) : 0ull;
      if (is_long) sum_l += add; else sum_s += add;
    }
    long_true = wave_sum(long_true);
    if ((threadIdx.x & (WAVE - 1)) == 0 && long_true) atomicAdd(&s_long_edges, long_true);
    u64 tot_s, tot_l;
    const u64 ex_s = block_exclusive_sum_lean<NW>(sum_s, s_scan, &tot_s);
    const u64 ex_l = block_exclusive_sum_lean<NW>(sum_l, s_scan, &tot_l);
    if (threadIdx.x == 0) {
      if (MGX_LAB_GET(a, build_diag, 0) & 4) { s_base[0] = ((u64)blockIdx.x * 4096) << BFS_VSHIFT; s_base[1] = ((u64)blockIdx.x * 4096) << BFS_VSHIFT; }
      else {
      s_base[0] = (tot_s >> 40) ? atomicAdd(cur_s, ((tot_s >> 40) << BFS_VSHIFT) | (tot_s & DEGMASK)) : 0ull;
      s_base[1] = (tot_l >> 40) ? atomicAdd(cur_l, ((tot_l >> 40) << BFS_VSHIFT) | (tot_l & DEGMASK)) : 0ull;
      }
      if (tot_l >> 40) atomicAdd(&c->ledges[(slot + 1) % 3], (u64)s_long_edges);     // (complete: two barriers ago)
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      if (dg[q] != 0u) {
        const bool is_long = (longmask >> q) & 1u;
        const u64 base = is_long ? s_base[1] : s_base[0];
        const u64 at = (is_long ? ex_l : ex_s) + loc[q];
        const u64 slot = (base >> BFS_VSHIFT) + (at >> 40);
        (is_long ? out_row_l : out_row_s)[slot] = ro[q];
        (is_long ? out_off_l : out_off_s)[slot] =
            (u32)((base & BFS_EMASK) + (at & DEGMASK)) | (is_long ? (dg[q] & 63u) : 0u);
      }
    }
    __syncthreads();       // st_v and s_base are reused by the next batch
  }
}

// ---- the same for the single-GPU path, without the detour through a list -------------------------------------------
// k_bfs_build above compacts a workgroup's discoveries into an LDS list and redistributes them over its threads before it
// touches their row extents and labels.  A thread's 16 vertices are CONSECUTIVE, though: row_offsets[i0 .. i0 + 16] and
// old_of_new[i0 .. i0 + 15] are two contiguous pieces (68 + 64 bytes per thread, 4 KB + 4 KB contiguous per wave), read
// with a few wide loads by every thread that found anything -- no gathers at all -- and everything a discovery needs is
// then in the registers of the thread that found it: degree, class, label target.  One packed workgroup scan per queue
// gives the positions; a thread writes its own discoveries (consecutive queue slots per class) and labels.  No list, no
// batches, no second distribution: three barriers' worth of scans instead of seven plus the list's.
// (Measured, RMAT-22, the level with 2 M discoveries: 46 us for the list version, 12.5 of them the label scatter.)
//
// LAZY queues.  The two bodies that read the unit blocks and the degree-sorted CSR (bfs_fused_dense.hpp,
// bfs_fused_vshort.hpp) take the frontier from frontier_bits: a slot that runs both never looks at its queues.  Which
// body a slot takes depends on the size of its frontier, which is only known when the build is over -- but the build can
// know how many marks the push stored (bfs_body_finish: one add per workgroup into the slot's counters): an upper bound
// of the discoveries, a few times their number on a skewed graph.  Behind a push that stored at least n / lazy_div marks
// (grid-uniform: the counters are complete when the build starts) the build writes NO queues: no positions, hence no
// scans and no returning atomics on the two hot cursors (500 workgroups queue up there at 12 ns each), no queue stores
// -- bits, labels and totals only (the cursors still receive the level's counts, with adds nobody waits for).
// ctrl->lazy_slot tells the next slot's push launch that it MUST take the queue-less bodies, whatever its size turns out
// to be (a sparse frontier then costs their sweeps and LDS copies: ~27 us instead of ~10 on RMAT-22, which is why the
// rule looks at the marks and not at the level's edges: few marks guarantee a small frontier).
// Measured, RMAT-22: the two big builds 41 -> 24 and 27 -> 20 us.
// (Round 5, tried and dropped: a lazy build that still writes the LONG-row queue -- the level behind the peak has a million short
//  rows and a few thousand long ones, and without a queue its long-row half sweeps every unit owner and runs the cold-edge pass for
//  them.  With the queue that half became a short walk, but the launch ends with its short-row half either way: push 183.4 -> 182.1
//  us per traversal, builds 89.9 -> 93.5 (the scan and the returning atomic behind the hub level), 0.3012 -> 0.3031 ms.)
__device__ __forceinline__ bool bfs_build_is_lazy(const bfs_fused_args_t& a, int slot) {
  if (a.lazy_div == 0u || !a.slot_marks) return false;
  u64 M = 0;
#pragma unroll
  for (int i = 0; i < BFS_MARK_CTRS; ++i) M += a.slot_marks[((slot & 1) * BFS_MARK_CTRS + i) * BFS_MARK_STRIDE];
  return M * (u64)a.lazy_div >= (u64)(u32)a.n;
}

// DIST (a rank of the partitioned traversal, bfs_dist2.hpp): the new vertices are the bits this rank OWNS of the level's merged
// discoveries (`bits`, over all vertices: local vertex i is global vertex i * ranks + rank) instead of marks; bitmap and
// frontier are k_d2_or's business; labels and row extents are local and contiguous: no scatter at all.
// DIST == 2: the same with the OR-merge of the ranks' new-bit maps INSIDE (what k_d2_or did in a launch of its own in front:
// ~12 us of launch and sweep per dense level).  With RANKS a power of two a thread's 16 local vertices are bits
// rank, rank + RANKS, ... of RANKS / 2 consecutive words of the global bitmap -- exactly the words no other thread looks at: the
// thread ORs those words of all maps, stores them as the level's frontier bitmap (`merged`), ORs them into its visited bitmap,
// clears them in the rank's own map (bfs_fused_sparse.hpp) and picks its 16 bits.  The grid covers the words, not only the rows.
struct bfs_d2_fuse_t {
  const u32* maps = nullptr;   // nmaps new-bit maps, stride words apart
  int nmaps = 0;
  long long stride = 0;
  long long nwords = 0;        // words of a map (a multiple of 4)
  u32* merged = nullptr;
  u32* clear = nullptr;        // the rank's own map (may be one of `maps`: a thread clears what it has read); NULL: none
  u32* list_head = nullptr;    // the rank's id list (its count is reset: nobody reads the list in a level merged from bitmaps); NULL: none
};
template <int NT, int DIST = 0, int RANKS = 1>
__global__ __launch_bounds__(NT, 4) void k_bfs_build2(bfs_fused_args_t a, int arg, int* __restrict__ labels, int n,
                                                      const u32* __restrict__ bits = nullptr, int ranks = 1, int rank = 0,
                                                      bfs_d2_fuse_t fz = bfs_d2_fuse_t()) {
  constexpr int NW = NT / WAVE;
  constexpr u64 CNT1 = 1ull << 40;
  constexpr u64 DEGMASK = CNT1 - 1ull;
  __shared__ u64 s_scan[NW + 1];
  __shared__ u64 s_base[2];
  __shared__ u32 s_long_edges;
  __shared__ u32 s_or[NW][WAVE];
  bfs_ctrl_t* const c = a.ctrl;
  // (round 6) the thread's 16 marks and its half-word of the bitmap are asked for BEFORE the control block is looked at: their
  // addresses follow from the thread's position alone, so the two round trips -- control block, then marks -- become one.  (A launch
  // that returns at once has read 18 bytes per thread for nothing; marks and bitmap are padded: always readable.)
  uint4 pre_m = make_uint4(0u, 0u, 0u, 0u);
  u32 pre_old16 = 0u;
  if (!DIST) {
    const long long j0 = ((((long long)blockIdx.x + (long long)(threadIdx.x % NW) * gridDim.x) * 64 + (threadIdx.x / NW)) * 16);
    const long long jc = j0 < (long long)n ? j0 : 0;
    pre_m = *(const uint4*)(a.mark + jc);
    pre_old16 = ((const unsigned short*)a.visited)[jc >> 4];
  }
  int slot, level;
  bfs_resolve(c, arg, slot, level);
  if (!DIST && (c->done || c->skip_build[slot & 3])) return;      // (a rank of a partitioned run may be handed work behind an empty frontier of its own)
  if (DIST && bfs_d2_frozen(c, level)) return;
  // (a level merged from bitmaps: the header of the rank's id list goes back to zero for the next level -- its push may append)
  if (DIST == 2 && fz.list_head && blockIdx.x == 0 && threadIdx.x == 0) fz.list_head[0] = 0u;
  // (direction-optimising runs: once the traversal has switched to bottom-up levels it stays there, and those read the
  //  frontier bitmap: no queue is ever needed again)
  const bool lazy = !DIST && (bfs_build_is_lazy(a, slot) || (a.lazy_pull && c->pull));
  if (!DIST && blockIdx.x == 0 && threadIdx.x == 0) {
    c->fb_slot = slot + 1;                                             // frontier_bits: written in full below
    c->lazy_slot = lazy ? slot + 1 : -1;
    if (lazy) c->lazy_slots += 1;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // A workgroup owns NW runs of 1024 vertices, gridDim runs apart (under the hub-first layout a level's discoveries
  // sit in a prefix of the ids: every workgroup gets its share of it).  Inside the workgroup a run is spread over ALL
  // waves -- thread t takes group t / NW (16 vertices) of run t % NW -- so that the one or two runs that hold the
  // workgroup's discoveries keep every wave busy instead of one; a wave still reads NW pieces of 128 contiguous bytes.
  const int my_run = threadIdx.x % NW, my_group = threadIdx.x / NW;
  const long long i0 = (((long long)blockIdx.x + (long long)my_run * gridDim.x) * 64 + my_group) * 16;
  const int new_label = level + 1;
  const u32 long_min = a.long_min > 0 ? (u32)a.long_min : 0xFFFFFFFFu;

  // deferred hot marks of the slot (bfs_hot_epilogue): the flush buffers' bits for this workgroup's runs.  A run is 128
  // bytes of every buffer: 16-byte loads, eight lanes per buffer, eight buffers per wave instruction, the F buffers
  // spread over the workgroup's waves -- all loads of a wave in flight at once (2-byte loads, one buffer per instruction:
  // the OR of a level that deferred in 512 workgroups took 10 us of its build).
  u32 flushed16 = 0;
  if (!DIST && a.flush_buf) {
    const u32 F = (MGX_LAB_GET(a, build_diag, 0) & 32) ? 0u : c->flush_count[slot & 1];     // (lab builds, 32: the OR of the flushed bitmaps skipped)
    if (F) {                                                     // (grid-uniform)
      for (int hr = 0; hr < NW; ++hr) {
        const long long run = (long long)blockIdx.x + (long long)hr * gridDim.x;
        if (run >= (long long)(a.defer_words / 32u)) break;
        const uint4* const base = (const uint4*)(a.flush_buf + (size_t)run * 32) + (lane & 7);
        uint4 acc = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll 8
        for (u32 k = (u32)wave * 8u + ((u32)lane >> 3); k < F; k += 8u * NW) {
          const uint4 v = base[(size_t)k * (BFS_FLUSH_WORDS / 4)];
          acc.x |= v.x; acc.y |= v.y; acc.z |= v.z; acc.w |= v.w;
        }
#pragma unroll
        for (int sh = 8; sh < 64; sh <<= 1) {
          acc.x |= __shfl_xor(acc.x, sh, WAVE); acc.y |= __shfl_xor(acc.y, sh, WAVE);
          acc.z |= __shfl_xor(acc.z, sh, WAVE); acc.w |= __shfl_xor(acc.w, sh, WAVE);
        }
        if (lane < 8) *(uint4*)(&s_or[wave][lane * 4]) = acc;
        __syncthreads();
        if (my_run == hr) {
#pragma unroll
          for (int w = 0; w < NW; ++w) flushed16 |= (u32)((const unsigned short*)&s_or[w][0])[my_group];
        }
        __syncthreads();
      }
    }
  }

  // the same for the bitmaps of the slot's cold-edge pass (bfs_fused_cold.hpp): cold workgroup k of slice s wrote what it
  // discovered in [cold_lo[s], + BFS_COLD_WORDS * 32) into buffer k.  Wave w takes the workgroup's run w: 16-byte loads,
  // eight lanes per buffer (a run is 128 bytes of bitmap), eight buffers per instruction, everything in flight at once.
  if (!DIST && a.cold_dst && c->cold_slot == slot && !(MGX_LAB_GET(a, build_diag, 0) & 64)) {                 // (grid-uniform; lab builds, 64: skipped)
    static_assert(NW == 8, "one wave per run");
    u32* const s_run = &s_or[0][0];                                  // [NW][32] words
    const long long run = (long long)blockIdx.x + (long long)wave * gridDim.x;
    const long long v0 = run * 1024;
    int sl = -1;
    for (int q = 0; q < a.cold_slices; ++q)
      if (v0 >= (long long)a.cold_lo[q] && v0 < (long long)a.cold_lo[q] + (long long)BFS_COLD_WORDS * 32) sl = q;
    uint4 acc = make_uint4(0u, 0u, 0u, 0u);
    if (sl >= 0) {                                                   // (wave-uniform)
      const u32 rel = (u32)((v0 - (long long)a.cold_lo[sl]) >> 10);
      const uint4* const base = (const uint4*)(a.cold_flush + (size_t)rel * 32) + (lane & 7);
      const u32 k1 = a.cold_wgs[sl + 1];
      for (u32 k = a.cold_wgs[sl] + ((u32)lane >> 3); k < k1; k += 8u) {
        const uint4 v = base[(size_t)k * (BFS_COLD_WORDS / 4)];
        acc.x |= v.x; acc.y |= v.y; acc.z |= v.z; acc.w |= v.w;
      }
#pragma unroll
      for (int sh = 8; sh < 64; sh <<= 1) {
        acc.x |= __shfl_xor(acc.x, sh, WAVE); acc.y |= __shfl_xor(acc.y, sh, WAVE);
        acc.z |= __shfl_xor(acc.z, sh, WAVE); acc.w |= __shfl_xor(acc.w, sh, WAVE);
      }
    }
    if (lane < 8) *(uint4*)(s_run + wave * 32 + lane * 4) = acc;
    __syncthreads();
    flushed16 |= (u32)((const unsigned short*)(s_run + my_run * 32))[my_group];
    __syncthreads();
  }

  // ---- which of my 16 vertices are new -------------------------------------------------------------------------------
  u32 new16 = 0;
  if constexpr (DIST == 2) {
    static_assert(RANKS >= 2 && RANKS <= 16 && (RANKS & (RANKS - 1)) == 0, "ranks: 2, 4, 8 or 16");
    constexpr int RW = RANKS / 2;                                            // words of the bitmap per thread
    __shared__ u32 s_found;
    if (threadIdx.x == 0) s_found = 0;
    if (blockIdx.x == 0 && threadIdx.x == 0) c->fb_slot = level + 1;         // merged = the frontier of level + 1 as a bitmap
    const long long w0 = (i0 >> 4) * RW;
    u32 g[RW];
#pragma unroll
    for (int k = 0; k < RW; ++k) g[k] = 0u;
    const bool in = w0 < fz.nwords;                                          // (RW <= 4: then all RW words are; RW == 8: see below)
    u32 found = 0;
    if (in) {
#pragma unroll 8
      for (int r = 0; r < fz.nmaps; ++r) {
        const u32* const mp = fz.maps + (size_t)r * (size_t)fz.stride + w0;
        if constexpr (RW >= 4) {
#pragma unroll
          for (int q4 = 0; q4 < RW / 4; ++q4) {
            if (q4 == 0 || w0 + 4 * q4 < fz.nwords) {
              const uint4 v = ((const uint4*)mp)[q4];
              g[4 * q4] |= v.x; g[4 * q4 + 1] |= v.y; g[4 * q4 + 2] |= v.z; g[4 * q4 + 3] |= v.w;
            }
          }
        } else if constexpr (RW == 2) {
          const uint2 v = *(const uint2*)mp;
          g[0] |= v.x; g[1] |= v.y;
        } else {
          g[0] |= mp[0];
        }
      }
#pragma unroll
      for (int k = 0; k < RW; ++k) {
        if (k < 4 || w0 + k < fz.nwords) {
          fz.merged[w0 + k] = g[k];
          if (fz.clear) fz.clear[w0 + k] = 0u;
          if (g[k]) a.visited[w0 + k] |= g[k];                               // (this thread is the word's only writer)
          found += (u32)__popc(g[k]);
        }
      }
    }
    // every rank counts the level's discoveries itself (ctrl->merged_new): one add per workgroup
    found = wave_sum(found);
    __syncthreads();                                                         // (s_found = 0 is in)
    if (lane == 0 && found) atomicAdd(&s_found, found);
    __syncthreads();
    if (threadIdx.x == 0 && s_found) atomicAdd(&c->merged_new, (u64)s_found);
    if (i0 < n) {
      const u32 valid = (n - i0 >= 16) ? 0xFFFFu : ((1u << (int)(n - i0)) - 1u);
#pragma unroll
      for (int q = 0; q < 16; ++q) new16 |= ((g[(q * RANKS) >> 5] >> (((q * RANKS) & 31) + rank)) & 1u) << q;
      new16 &= valid;
    }
  } else if (DIST) {
    if (i0 < n) {
      const u32 valid = (n - i0 >= 16) ? 0xFFFFu : ((1u << (int)(n - i0)) - 1u);
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const long long v = (i0 + q) * (long long)ranks + rank;              // (past the rank's last row: masked by `valid`; the bitmap is padded)
        if (((valid >> q) & 1u) && ((bits[v >> 5] >> (v & 31)) & 1u)) new16 |= 1u << q;
      }
    }
  } else if (i0 < n) {
    const u32 valid = (n - i0 >= 16) ? 0xFFFFu : ((1u << (int)(n - i0)) - 1u);
    const uint4 m = pre_m;                                                   // (loaded at the kernel's entry)
    const u32 x[4] = {m.x, m.y, m.z, m.w};
    u32 m16 = flushed16;
#pragma unroll
    for (int q = 0; q < 4; ++q) m16 |= (((x[q] & 0x01010101u) * 0x10204080u) >> 28) << (4 * q);   // 4 marks -> 4 bits
    unsigned short* const vis16 = (unsigned short*)a.visited + (i0 >> 4);
    const u32 old16 = pre_old16;
    new16 = m16 & ~old16 & valid;
    if (new16) *vis16 = (unsigned short)(old16 | new16);                     // this thread is the half-word's only writer
    ((unsigned short*)a.frontier_bits)[i0 >> 4] = (unsigned short)new16;     // bottom-up levels, unit blocks
  }
  if (DIST && a.d2_front && i0 < n) ((unsigned short*)a.d2_front)[i0 >> 4] = (unsigned short)new16;     // (the rank's short rows, vertex by vertex)
  const int mine = __popc(new16);

  // ---- my discoveries' row extents and label targets: contiguous, no gathers ----------------------------------------
  u32 ro[17];
  int target[16];
  u64 sum_s = 0, sum_l = 0;
  u32 long_true = 0;
  const bool diag_extents = (MGX_LAB_GET(a, build_diag, 0) & 16) != 0;     // (lab builds: synthetic extents)
  if (diag_extents && new16) {
#pragma unroll
    for (int q = 0; q < 17; ++q) ro[q] = (u32)(i0 + q) * 3u;
#pragma unroll
    for (int q = 0; q < 16; ++q) target[q] = (int)(i0 + q);
  } else if (new16) {
    if (i0 + 16 < (long long)n) {                                            // (row_offsets has n + 1 entries)
      const u32* const p = a.row_offsets + i0;                               // i0 is a multiple of 16: 64-byte aligned
      const uint4 r0 = *(const uint4*)p, r1 = *(const uint4*)(p + 4), r2 = *(const uint4*)(p + 8), r3 = *(const uint4*)(p + 12);
      ro[0] = r0.x; ro[1] = r0.y; ro[2] = r0.z; ro[3] = r0.w; ro[4] = r1.x; ro[5] = r1.y; ro[6] = r1.z; ro[7] = r1.w;
      ro[8] = r2.x; ro[9] = r2.y; ro[10] = r2.z; ro[11] = r2.w; ro[12] = r3.x; ro[13] = r3.y; ro[14] = r3.z; ro[15] = r3.w;
      ro[16] = p[16];
    } else {
#pragma unroll
      for (int q = 0; q < 17; ++q) ro[q] = a.row_offsets[(i0 + q <= (long long)n) ? i0 + q : (long long)n];
    }
    if (a.old_of_new) {
      if (i0 + 16 <= (long long)n) {
        const int* const p = a.old_of_new + i0;
        const int4 t0 = *(const int4*)p, t1 = *(const int4*)(p + 4), t2 = *(const int4*)(p + 8), t3 = *(const int4*)(p + 12);
        target[0] = t0.x; target[1] = t0.y; target[2] = t0.z; target[3] = t0.w; target[4] = t1.x; target[5] = t1.y; target[6] = t1.z; target[7] = t1.w;
        target[8] = t2.x; target[9] = t2.y; target[10] = t2.z; target[11] = t2.w; target[12] = t3.x; target[13] = t3.y; target[14] = t3.z; target[15] = t3.w;
      } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) target[q] = a.old_of_new[(i0 + q < (long long)n) ? i0 + q : (long long)n - 1];
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) target[q] = (int)(i0 + q);
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      if ((new16 >> q) & 1u) {
        if (!(MGX_LAB_GET(a, build_diag, 0) & 1)) labels[target[q]] = new_label;
        const u32 deg = ro[q + 1] - ro[q];
        if (deg >= long_min) { sum_l += CNT1 | (u64)bfs_lq_pad(deg); long_true += deg; }
        else if (deg) sum_s += CNT1 | (u64)deg;
      }
    }
  }

  if (lazy) {
    // totals only: (count << 40 | edges) per queue, true long edges, discoveries -- wave sums, one add per wave into LDS,
    // one add per workgroup and counter into the control block (no return value: nobody waits on the hot lines)
    const u64 w_s = wave_sum(sum_s), w_l = wave_sum(sum_l);
    const u32 w_t = wave_sum(long_true), w_n = wave_sum((u32)mine);
    if (threadIdx.x < 4) s_scan[threadIdx.x] = 0;
    __syncthreads();
    if (lane == 0 && w_n) {
      atomicAdd(&s_scan[0], w_s); atomicAdd(&s_scan[1], w_l); atomicAdd(&s_scan[2], (u64)w_t); atomicAdd(&s_scan[3], (u64)w_n);
    }
    __syncthreads();
    if (threadIdx.x == 0 && s_scan[3]) {
      const u64 tot_s = s_scan[0], tot_l = s_scan[1];
      atomicAdd(&c->reached, s_scan[3]);
      if (tot_s >> 40) atomicAdd(&c->cursor[(slot + 1) % 3], ((tot_s >> 40) << BFS_VSHIFT) | (tot_s & DEGMASK));
      if (tot_l >> 40) {
        atomicAdd(&c->lcursor[(slot + 1) % 3], ((tot_l >> 40) << BFS_VSHIFT) | (tot_l & DEGMASK));
        atomicAdd(&c->ledges[(slot + 1) % 3], s_scan[2]);
      }
    }
    return;
  }
  // ---- positions: one packed scan per queue; batches of a workgroup are appended with one atomic per queue -------------
  if (threadIdx.x == 0) s_long_edges = 0;
  u64 tot_n;
  (void)block_exclusive_sum_lean<NW>((u64)mine, s_scan, &tot_n);             // (also orders s_long_edges = 0 before the adds)
  if (tot_n == 0) return;
  long_true = wave_sum(long_true);
  if (lane == 0 && long_true) atomicAdd(&s_long_edges, long_true);
  u64 tot_s, tot_l;
  u64 ex_s = block_exclusive_sum_lean<NW>(sum_s, s_scan, &tot_s);
  u64 ex_l = block_exclusive_sum_lean<NW>(sum_l, s_scan, &tot_l);
  // the two returning atomics from two different waves: one thread would wait for the first before it issues the second
  if (threadIdx.x == 0) {
    atomicAdd(&c->reached, tot_n);
    if (MGX_LAB_GET(a, build_diag, 0) & 4) s_base[0] = ((u64)blockIdx.x * 8192) << BFS_VSHIFT;
    else s_base[0] = (tot_s >> 40) ? atomicAdd(&c->cursor[(slot + 1) % 3], ((tot_s >> 40) << BFS_VSHIFT) | (tot_s & DEGMASK)) : 0ull;
  }
  if (threadIdx.x == WAVE) {
    if (MGX_LAB_GET(a, build_diag, 0) & 4) s_base[1] = ((u64)blockIdx.x * 8192) << BFS_VSHIFT;
    else {
      s_base[1] = (tot_l >> 40) ? atomicAdd(&c->lcursor[(slot + 1) % 3], ((tot_l >> 40) << BFS_VSHIFT) | (tot_l & DEGMASK)) : 0ull;
      if (tot_l >> 40) atomicAdd(&c->ledges[(slot + 1) % 3], (u64)s_long_edges);
    }
  }
  __syncthreads();
  if (!new16 || (MGX_LAB_GET(a, build_diag, 0) & 8)) return;
  u32* __restrict__ const out_row_s = a.fr_row[(slot + 1) & 1];
  u32* __restrict__ const out_off_s = a.fr_off[(slot + 1) & 1];
  u32* __restrict__ const out_row_l = a.lq_row[(slot + 1) & 1];
  u32* __restrict__ const out_off_l = a.lq_off[(slot + 1) & 1];
  const u64 base_s = s_base[0], base_l = s_base[1];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    if ((new16 >> q) & 1u) {
      const u32 deg = ro[q + 1] - ro[q];
      if (deg >= long_min) {
        const u64 at = (base_l >> BFS_VSHIFT) + (ex_l >> 40);
        out_row_l[at] = ro[q];
        out_off_l[at] = (u32)((base_l & BFS_EMASK) + (ex_l & DEGMASK)) | (deg & 63u);
        ex_l += CNT1 | (u64)bfs_lq_pad(deg);
      } else if (deg) {
        const u64 at = (base_s >> BFS_VSHIFT) + (ex_s >> 40);
        out_row_s[at] = ro[q];
        out_off_s[at] = (u32)((base_s & BFS_EMASK) + (ex_s & DEGMASK));
        ex_s += CNT1 | (u64)deg;
      }
    }
  }
}

// End of a batch of launches: the control block's head goes to the host's pinned copy, then a sequence number the host
// is spinning on.  Instead of a D2H copy (a blit kernel of its own, ~4.5 us) followed by hipStreamSynchronize (the
// runtime's wake-up path): the words are stored straight into host memory by one wave, fenced at system scope, and the
// host reads them the moment the number changes.
__global__ void k_bfs_publish(const bfs_ctrl_t* __restrict__ c, bfs_ctrl_t* __restrict__ host, u64* host_seq, u64 seq, int words) {
  const u32* const src = (const u32*)c;
  u32* const dst = (u32*)host;
  for (int i = threadIdx.x; i < words; i += blockDim.x) dst[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(host_seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Tuning switches of the fused traversal, read from the environment ONCE per handle (when its state is made -- the
// parity tests set them and then create the problem), never per traversal or per launch.  Every one of them selects
// between product paths that the default thresholds pick by size; the shapes that lost their A/B runs and the
// instrumented kernels exist in lab builds only (-DMGX_LAB).
struct bfs_run_opts_t {
  int cold_test = -1;      // MGX_BFS_COLD_TEST
  int merged = 1;          // MGX_BFS_MERGED_PUSH
  int dense = -1;          // MGX_BFS_DENSE: 0 never, N > 0 dense_div = N
  int vshort = -1;         // MGX_BFS_VSHORT: 0 never, N > 0 vshort_div = N
  long long chain = -1;    // MGX_BFS_CHAIN_MAX_EDGES: 0 never (default BFS_CHAIN_CAP)
  int pack24 = 1;                     // MGX_BFS_PACK24=0
  int hot_units = 1;                  // MGX_BFS_HOT_UNITS=0: the full unit blocks even when the layout carries the ones without the cold entries
  int cold_pack = 1;                  // MGX_BFS_COLD_PACK=0: the cold-edge pass reads its pairs at eight bytes each: the unit-block body reads the 32-bit entries even when the graph carries the 24-bit copy
  int src_plan = 1;                   // MGX_BFS_SRC_PLAN=0: every traversal gets the same launch sequence (no per-source classes)
  int defer_words = -1;               // MGX_BFS_DEFER_WORDS: words of the bitmap prefix whose marks are deferred (default: all BFS_FLUSH_WORDS)
  int defer_mul = 1, defer_div = 1;   // MGX_BFS_DEFER_REACH="mul/div": a level defers its hot marks while reached * mul < range * div
                                      // (RMAT-22, ms per traversal: 4/1 0.3627, 2/1 0.3600, 1/1 0.3521, 2/3 0.3511, 1/3 0.3530, 1/8 0.3625, always 0.3641)
  int do_chain = 1;        // MGX_BFS_DO_CHAIN=0: direction-optimising runs keep every level device-wide (no chains of small top-down levels)
  int merged_pull = 1;     // MGX_BFS_MERGED_PULL=0: the bottom-up sweep as a launch of its own behind every push launch
  int seed_chain = 1;      // MGX_BFS_SEED_CHAIN=0: no in-place chain launches at all (small levels run inside the slots' push launches)
  int tail_chain = 1;      // MGX_BFS_TAIL_CHAIN=0: ... only the one at the start
  int tail_front = 1;      // MGX_BFS_TAIL_FRONT=0: no chain launch in FRONT of the last slots of a batch (only behind it)
  int chain_big = -1;      // MGX_BFS_CHAIN_BIG_EDGES: largest level of an in-place chain launch
  int cold = 1;            // MGX_BFS_COLD: 0 the unit-block body marks its cold entries itself (no cold-edge pass), 2 the long rows' lists only,
                           // 1 (default) also the short rows' where the layout carries them (graphs of more than 2^23 vertices; equal on RMAT-22:
                           // 0.3712 / 0.3708 ms in round 4)
  int lazy = -1;           // MGX_BFS_LAZY: 0 the queue build always writes the queues, N: not behind a push that stored >= n / N marks
  long long defer = -1;    // MGX_BFS_DEFER: 0 never defer hot marks, N: flush a bitmap above N deferred marks per workgroup
  int spin = -1;           // (no switch since round 6) 0 read the control block back with a copy + hipStreamSynchronize, 1 publish kernel + spin, -1 the handle's default
  int build_list = 0;      // MGX_BFS_BUILD_LIST=1: the list-based queue build (k_bfs_build) instead of k_bfs_build2
  int mini = 1;            // MGX_BFS_MINI=0: no M launches (mid-size levels take device-wide slots; bfs_fused_mini.hpp); 1: on graphs of at
                           // least 2^22 vertices; 2: always
  int many_spare = 0;      // (no switch since round 6: re-runs teach the handle, auto_spare) launch slots a traversal of a batch (mgx_bfs_run_many) gets beyond what the last
                           // traversals of the graph needed (the most any of the last four needed: one that still does not finish is run
                           // again on its own, and the chain behind a traversal's last slot takes stragglers of up to BFS_CHAIN_CAP_BIG edges)
#ifdef MGX_LAB
  int flags = 0;           // MGX_BFS_FLAGS (instrumented stream kernel)
  int build_diag = 0;      // MGX_BFS_BUILD_DIAG: parts of k_bfs_build switched off (measurements only)
  int dense_diag = 0;      // MGX_BFS_DENSE_DIAG: parts of the unit-block body switched off (measurements only)
#endif
  static bfs_run_opts_t from_env() {
    bfs_run_opts_t o;
    auto geti = [](const char* name, int& out) { if (const char* e = mgx::env(name)) out = atoi(e); };
    auto getll = [](const char* name, long long& out) { if (const char* e = mgx::env(name)) out = atoll(e); };
    geti("MGX_BFS_COLD_TEST", o.cold_test);
    geti("MGX_BFS_MERGED_PUSH", o.merged);
    geti("MGX_BFS_DENSE", o.dense);
    geti("MGX_BFS_VSHORT", o.vshort);
    getll("MGX_BFS_CHAIN_MAX_EDGES", o.chain);
    geti("MGX_BFS_BUILD_LIST", o.build_list);
    getll("MGX_BFS_DEFER", o.defer);
    geti("MGX_BFS_DEFER_WORDS", o.defer_words);
    geti("MGX_BFS_SRC_PLAN", o.src_plan);
    geti("MGX_BFS_PACK24", o.pack24);
    geti("MGX_BFS_COLD_PACK", o.cold_pack);
    geti("MGX_BFS_HOT_UNITS", o.hot_units);
    geti("MGX_BFS_COLD", o.cold);
    if (const char* e = mgx::env("MGX_BFS_DEFER_REACH")) {
      o.defer_mul = atoi(e);
      const char* sl = strchr(e, '/');
      o.defer_div = sl ? atoi(sl + 1) : 1;
      if (o.defer_div < 1) o.defer_div = 1;
    }
    geti("MGX_BFS_SEED_CHAIN", o.seed_chain);
    geti("MGX_BFS_MERGED_PULL", o.merged_pull);
    geti("MGX_BFS_DO_CHAIN", o.do_chain);
    geti("MGX_BFS_TAIL_CHAIN", o.tail_chain);
    geti("MGX_BFS_TAIL_FRONT", o.tail_front);
    geti("MGX_BFS_CHAIN_BIG_EDGES", o.chain_big);
    geti("MGX_BFS_MINI", o.mini);
    if (o.many_spare < 0) o.many_spare = 0;
    geti("MGX_BFS_LAZY", o.lazy);
    if (o.lazy > (1 << 20)) o.lazy = 1 << 20;     // (edges < 2^38: no overflow)
#ifdef MGX_LAB
    geti("MGX_BFS_FLAGS", o.flags);
    geti("MGX_BFS_BUILD_DIAG", o.build_diag);
    geti("MGX_BFS_DENSE_DIAG", o.dense_diag);
#else
#endif
    return o;
  }
};

// per-BFS device state of the fused engine
struct bfs_fused_state_t {
  bfs_run_opts_t opts = bfs_run_opts_t::from_env();
  mem_t<u32> visited;
  mem_t<unsigned char> mark;
  mem_t<u32> frontier_bits;
  mem_t<u32> fr_row[2];
  mem_t<u32> fr_off[2];
  mem_t<u32> lq_row[2];
  mem_t<u32> lq_off[2];
  mem_t<bfs_ctrl_t> ctrl;
  mem_t<u32> flush_buf;              // deferred hot marks: BFS_FLUSH_MAX bitmaps of BFS_FLUSH_WORDS words (allocated on demand)
  unsigned defer_min_marks = 2048;   // a push workgroup with more deferred discoveries flushes a bitmap (MGX_BFS_DEFER: 0 = never defer)
  bfs_ctrl_t* host_ctrl = nullptr;   // pinned copy for stats
  u64* host_seq = nullptr;           // pinned: the batch number k_bfs_publish stores when the copy above is complete
  u64 seq = 0;
  bool spin = true;                  // wait for a batch by spinning on host_seq (false: copy + hipStreamSynchronize)
  int n = 0;
  int levels_per_sync = 2;           // slots (bfs_fused_run.hpp) launched between two read-backs of the control block ...
  int slots_hint = 5;                // ... except for the first batch: as many slots as the previous traversal needed
  int slots_used = 0;                // slots the last run launched
  bool time_batches = false;         // HIP events around every batch of launches (-> level_kernel_ms; ~6 us each)
  unsigned chain_max_edges = 4096;   // largest level block 0 of a push launch runs itself, chained with the small levels behind it
                                     // (bfs_fused_chain.hpp; <= BFS_CHAIN_CAP; early in a traversal BFS_CHAIN_EARLY_EDGES; 0: never)
  unsigned vshort_div = 8;           // short rows are walked vertex by vertex when the level holds at least 1 / vshort_div of
                                     // all short-row edges (bfs_fused_vshort.hpp; 0: never)
  unsigned chain_big_edges = 4096;   // largest level the in-place chain kernel runs (bfs_fused_run.hpp; <= BFS_CHAIN_CAP_BIG): a lone workgroup
                                     // needs ~5 + 4.3 us per 1000 edges (measured: 10 487 edges 49 us, 7 391 38 us, 1 281 17 us), a device-wide slot ~21 us
  int recent_need[4] = {1, 1, 1, 1}, recent_at = 0;   // slots the last traversals needed
  // ... and per source class (bfs_classify_source, bfs_fused_run.hpp: 1 = the M launch in front absorbs the first level the
  // chain leaves, 2 = that level is too big for it and the launch is not enqueued): the slots the last four traversals of
  // the class needed under ITS launch sequence; a class without history takes the graph's hint
  int cls_max[3] = {0, 0, 0}, cls_at[3] = {0, 0, 0};   // the most any traversal of the class has needed on this handle (never shrinks: a
                                                        // class whose members differ -- RMAT-24: 3 or 4 slots -- must not oscillate into re-runs)
  int auto_spare = 0, clean_batches = 0;             // batches (bfs_fused_run_many): spare slots learnt from re-runs, batches without one since
  int tail_from = 1 << 30;           // slots from this one on get an in-place chain launch in front (learnt from the previous traversal)
  unsigned lazy_div = 4;             // the build behind a push with >= n / lazy_div mark stores writes no queues (bfs_build_is_lazy; 0: never)
  mem_t<u32> slot_marks;             // the counters it looks at (bfs_fused_args_t::slot_marks)
  mem_t<u32> cold_flush;             // cold-edge pass: one bitmap of BFS_COLD_WORDS words per cold workgroup (allocated on demand)
  unsigned dense_div = 2;            // long rows are read from the unit blocks when the frontier holds at least
                                     // 1 / dense_div of the units (bfs_fused_dense.hpp; 0: never)
  int long_min = LONG_MIN_DEFAULT;   // rows at least this long go to the long-row queue (0: no such queue)
  bool count_marks = false;          // see bfs_fused_args_t::count_marks (mgx_bfs_set_kernel_timing switches it on)
  int time_kernels = 0;              // 1: the push parts as separate launches, HIP events around each; 2: events around the
                                     // ONE merged push launch of every slot (the product kernel; -> stream_kernel_ms) (each event
                                     // leaves a ~6 us gap on the stream: profiling runs only)
  unsigned hot_min_edges = 65536;    // smaller levels probe the bitmap in L2 instead of copying its hot prefix to LDS
  // timing of the level kernels of the last run (HIP events around each batch of launches)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  double level_kernel_ms = 0.0;
  long long level_kernel_launches = 0;
  float batch_ms[256];               // duration of each launch batch of the last run (per level when levels_per_sync == 1)
  int batches = 0;
  // per-launch timing of the two push kernels of a level: events [3i] stream [3i+1] wave [3i+2]
  static constexpr int EV_POOL = 96;
  hipEvent_t wev[EV_POOL] = {};
  double wave_kernel_ms = 0.0;       // the short-row part of k_bfs_push (timing mode 1)
  long long wave_kernel_launches = 0;
  double stream_kernel_ms = 0.0;     // the long-row part of k_bfs_push (timing mode 1) or the whole merged launch (mode 2)
  long long stream_kernel_launches = 0;
  float level_stream_ms[64] = {};    // the same per slot (first 64 slots)
  float level_wave_ms[64] = {};

  bfs_fused_state_t(int num_nodes, standard_context_t& ctx) : n(num_nodes) {
    size_t words = (size_t)(num_nodes + 31) / 32 + 8;
    if (words < 65536) words = 65536;              // the kernels copy a fixed-size prefix of the bitmap into LDS
    visited = mem_t<u32>(words, ctx);
    frontier_bits = mem_t<u32>(words, ctx);
    mark = mem_t<unsigned char>((size_t)num_nodes + 64, ctx);
    MGX_HIP(hipMemsetAsync(visited.data(), 0, words * sizeof(u32), ctx.stream()));
    MGX_HIP(hipMemsetAsync(frontier_bits.data(), 0, words * sizeof(u32), ctx.stream()));
    for (int i = 0; i < 2; ++i) {
      fr_row[i] = mem_t<u32>((size_t)num_nodes + 1, ctx);
      fr_off[i] = mem_t<u32>((size_t)num_nodes + 1, ctx);
      lq_row[i] = mem_t<u32>((size_t)num_nodes + 1, ctx);
      lq_off[i] = mem_t<u32>((size_t)num_nodes + 1, ctx);
    }
    ctrl = mem_t<bfs_ctrl_t>(1, ctx);
    MGX_HIP(hipHostMalloc((void**)&host_ctrl, sizeof(bfs_ctrl_t), hipHostMallocDefault));
    MGX_HIP(hipHostMalloc((void**)&host_seq, 64, hipHostMallocDefault));
    *host_seq = 0;
    MGX_HIP(hipEventCreate(&ev0));
    MGX_HIP(hipEventCreate(&ev1));
    for (int i = 0; i < EV_POOL; ++i) MGX_HIP(hipEventCreate(&wev[i]));
    if (const char* e = mgx::env("MGX_BFS_LONG_MIN")) long_min = atoi(e) > 0 ? atoi(e) : 0;
    if (const char* e = mgx::env("MGX_BFS_HOT_MIN_EDGES")) hot_min_edges = (unsigned)atoll(e);
  }
  bfs_fused_state_t(const bfs_fused_state_t&) = delete;
  bfs_fused_state_t& operator=(const bfs_fused_state_t&) = delete;
  ~bfs_fused_state_t() {
    if (host_ctrl) (void)hipHostFree(host_ctrl);
    if (host_seq) (void)hipHostFree(host_seq);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    for (int i = 0; i < EV_POOL; ++i) if (wev[i]) (void)hipEventDestroy(wev[i]);
  }
  size_t bitmap_words() const { return visited.size(); }
};

}  // namespace mgx
