// mgx/bfs_fused_run.hpp -- host driver of the fused BFS.  One init kernel, an in-place chain launch for the small levels
// at the start (k_bfs_chain_inplace: one workgroup), then per launch SLOT
//   k_bfs_push   ONE grid: [cold-edge pass | long rows: unit blocks or queue walk | short rows: vertex by vertex or searched]
//                -- or the bottom-up sweep of a direction-optimising run, or block 0 running a chain of small levels
//   k_bfs_build2 marks (+ flushed bitmaps) -> bitmap bits, labels, frontier bitmap, the next slot's queues (or none: lazy)
// enqueued back to back for as many slots as the last traversals of the graph needed, another in-place chain launch for
// the stragglers, and k_bfs_publish, on which the host spins: one host wait per batch.  A slot works on the level
// ctrl->slot_level[slot & 3].  RMAT-22: init + chain + 4 slots + chain + publish = 12 launches for 7 levels.
// The partitioned path (bfs_dist2.hpp) drives the queue-walk bodies with explicit level numbers (k_bfs_push_level).
#pragma once
#include <cstring>
#include <vector>
#include <unistd.h>
#include "bfs_fused.hpp"
#include "bfs_fused_chain.hpp"
#include "bfs_fused_cold.hpp"
#include "bfs_fused_dense.hpp"
#include "bfs_fused_mini.hpp"
#include "bfs_fused_pull.hpp"
#include "bfs_fused_stream.hpp"
#include "bfs_fused_vshort.hpp"
#include "bfs_fused_sparse.hpp"
#include "bfs_fused_wave.hpp"

namespace mgx {

// layout (optional): a hub-first relabelled copy of the CSR plus the two id maps; labels stay in the
// original id space either way.  Unit blocks (optional, of the same CSR): see bfs_fused_dense.hpp.
struct bfs_layout_t {
  const int* row_offsets = nullptr;
  const int* col_indices = nullptr;
  const int* new_of_old = nullptr;
  const int* old_of_new = nullptr;
  const int* ub_col = nullptr;
  const unsigned* ub_col24 = nullptr;   // the same entries, 24 bits each (graphs of at most 2^23 vertices; NULL: none)
  const int* ub_owner = nullptr;
  long long ub_units = 0, ub_units_pad = 0;
  int ub_min_degree = 0;            // the rows the unit blocks hold: degree >= this (must equal the long-row threshold)
  // the same rows without the entries of the cold-edge lists, 24 bits per entry (any graph size), their own owners: read instead of the
  // blocks above by a run that takes the cold-edge pass on every unit-block level (NULL: none)
  const unsigned* ubh_col24 = nullptr;
  const int* ubh_owner = nullptr;
  long long ubh_units = 0, ubh_units_pad = 0;
  // short rows vertex by vertex (bfs_fused_vshort.hpp): class boundaries of the degree-sorted CSR, edges of the range,
  // the long-row threshold they were computed for, index of four -1 behind col_indices (0: not available)
  unsigned vs_v[4] = {0, 0, 0, 0};
  unsigned vs_v9 = 0;               // first vertex of degree < 9 (0: unknown -- degrees 5 .. 16 are one class)
  unsigned vs_edges = 0, vs_dummy = 0;
  int vs_long_min = 0;
  // cold-edge lists of the long rows (bfs_fused_cold.hpp): pairs grouped by slice; hot_n / long_min they were cut for
  const int* cold_owner = nullptr;
  const int* cold_dst = nullptr;
  bool cold_pairs8 = false;             // the two arrays above exist (round 6: a layout whose slices are ALL packed drops them)
  const unsigned* cold_pk = nullptr;    // the same pairs, four bytes each (bfs_fused_args_t::cold_pk); NULL: none
  const unsigned* cold_cbase = nullptr;
  unsigned cold_cb[BFS_COLD_MAX_SLICES + 1] = {0};
  unsigned long long cold_pk_mask = 0;
  const int* colds_owner = nullptr;   // the short rows' cold entries (graphs of more than 2^23 vertices)
  const int* colds_dst = nullptr;
  int cold_slices = 0;
  unsigned cold_lo[BFS_COLD_MAX_SLICES] = {0}, cold_off[BFS_COLD_MAX_SLICES + 1] = {0}, colds_off[BFS_COLD_MAX_SLICES + 1] = {0},
           cold_wgs[BFS_COLD_MAX_SLICES + 1] = {0};
  unsigned cold_hot_n = 0;
  int cold_long_min = 0;
  bool cold_majority = false;         // more than a quarter of the long rows' entries point behind the LDS prefix (no lists were built): a flat graph
  bool cold_all = false;              // (round 6) a flat graph WITH lists: every entry of every row, slices from vertex 0 on (cold_hot_n == 0); cold_pairs_total of them
  unsigned long long cold_pairs_total = 0;
  // HOST table, 4 words per SOURCE OF THE CALL (entry i belongs to the i-th source handed to bfs_fused_run / _run_many; round 6: resolved
  // on demand, mgx/src_shapes.hpp): what a traversal from it starts with
  const unsigned* src_shapes = nullptr;
  int src_shapes_long_min = 0;
};

// A traversal's first levels, known before anything is enqueued (bfs_layout_t::src_shapes): level 0 is the source's row,
// level 1 its distinct neighbours.  The launch sequence of a traversal is [init][chain][M] slots ... [M][chain]; whether
// the M launch in FRONT finds work is decided by the first level the in-place chain leaves behind -- and for most sources
// that is level 0 or 1:
//   BFS_SRC_ABSORB  that level is mid-size: the M launch expands it, the device-wide slots start one level later;
//   BFS_SRC_SKIP    it is too big for an M launch, which would only forward it (a queue copy of 5-14 us on RMAT-22): the
//                   launch is not enqueued, slot 0 is the first device-wide slot;
//   BFS_SRC_UNKNOWN levels 0 and 1 are both small enough for the chain (or there is no table): the graph-wide sequence.
// Each class learns its own number of device-wide slots (bfs_fused_state_t::cls_need).  RMAT-22, 64 bench sources: 45 %
// ABSORB -- three slots instead of the four the other 55 % need, so their stragglers go to the M launch behind the slots
// instead of a device-wide slot with a queue build behind it -- and 55 % SKIP.  The rules mirror the device's own
// (bfs_level_is_chained with the in-place limits, bfs_level_is_mini) on exact numbers; a graph whose rows are not sorted
// by neighbour overestimates level 1 (duplicates count twice), which can only turn ABSORB into SKIP or UNKNOWN -- a slot
// more than needed, never one too few.
constexpr int BFS_SRC_UNKNOWN = 0, BFS_SRC_ABSORB = 1, BFS_SRC_SKIP = 2;

constexpr int BFS_STREAM_HOTW2 = 20400;   // two workgroups per CU: 80 KB of bitmap each
constexpr int BFS_WAVE_HOTW = 18000;
constexpr int BFS_DENSE_HOTW = BFS_STREAM_HOTW2 - 16;   // (the dense body's two sentinel words fit the same 81 664 bytes)
static_assert(BFS_COLD_WORDS == BFS_DENSE_HOTW, "a cold slice is as long as the unit-block body's LDS prefix");

constexpr size_t bfs_push_lds_bytes() {
  size_t m = bfs_stream_lds_bytes(BFS_STREAM_HOTW2);
  if (bfs_wave_lds_bytes(1024, BFS_WAVE_HOTW) > m) m = bfs_wave_lds_bytes(1024, BFS_WAVE_HOTW);
  if (bfs_dense_lds_bytes(BFS_DENSE_HOTW) > m) m = bfs_dense_lds_bytes(BFS_DENSE_HOTW);
  if (bfs_chain_lds_bytes() > m) m = bfs_chain_lds_bytes();
  if (bfs_vshort_lds_bytes(BFS_DENSE_HOTW) > m) m = bfs_vshort_lds_bytes(BFS_DENSE_HOTW);
  if (bfs_cold_lds_bytes() > m) m = bfs_cold_lds_bytes();
  return m;
}

// what a slot's push launch does, derived by every workgroup from the same stable inputs
struct bfs_slot_plan_t {
  int slot, level;
  bool empty, chained, dense, vshort, cold, colds, pulls;
};
__device__ __forceinline__ bfs_slot_plan_t bfs_slot_plan(const bfs_fused_args_t& a, int arg) {
  const bfs_ctrl_t* const c = a.ctrl;
  bfs_slot_plan_t p;
  bfs_resolve(c, arg, p.slot, p.level);
  const u64 cur = c->cursor[p.slot % 3], lcur = c->lcursor[p.slot % 3], ledges = c->ledges[p.slot % 3];
  p.empty = ((cur | lcur) >> BFS_VSHIFT) == 0 || c->done;
  p.chained = !p.empty && bfs_level_is_chained(a, cur, lcur, ledges);
  const bool pulls = bfs_level_pulls(a, c, p.slot);
  p.pulls = !p.empty && pulls;
  p.dense = !p.empty && !p.chained && !pulls && bfs_long_is_dense(a, c, p.slot, lcur);
  p.vshort = !p.empty && !p.chained && !pulls && bfs_short_is_dense(a, c, p.slot, cur);
  // the long rows' cold entries go through the pair lists (bfs_fused_cold.hpp) when the frontier holds enough long rows to
  // read the unit blocks by its own size (a sweep of all pairs does not pay for a sparse frontier that was only forced
  // onto the unit blocks by a lazy build: the unit-block body marks the few cold entries it meets -- in the FULL blocks: the ones
  // without the lists' entries, args.ub_hot_only, are for the levels that run the pass, bfs_dense_body)
  p.cold = p.dense && a.cold_dst != nullptr;
  // (lab builds: with them, the short rows' cold entries of a level that walks those vertex by vertex)
  p.colds = p.cold && p.vshort && a.colds_dst != nullptr;
  if (!p.empty && c->lazy_slot == p.slot) {       // the build before this slot wrote no queues (bfs_build_is_lazy)
    p.chained = false;
    p.dense = p.vshort = true;
  }
  if (a.cold_all) {
    // a flat graph: a level that holds an eighth of all entries (and whose frontier the build left as a bitmap) is ONE sweep of the
    // all-entries lists by the cold workgroups; any other level walks its queues (no unit blocks, no vertex-by-vertex walk: their
    // entries live in the lists)
    const u64 E = (cur & BFS_EMASK) + ledges;
    p.dense = p.vshort = p.colds = false;
    p.cold = !p.empty && !p.chained && !pulls && c->fb_slot == p.slot && E * 8ull >= a.cold_all_pairs;
  }
  return p;
}

// The opener of a device-wide level (one thread of the push launch): the level's bookkeeping, and the next slot's level.
// Nothing it writes is read by the slot's own push kernels: they take the queue sizes from the ring entry of the
// slot, which the previous launch completed.
__device__ __forceinline__ void bfs_slot_open(const bfs_fused_args_t& a, const bfs_slot_plan_t& p) {
  bfs_ctrl_t* const c = a.ctrl;
  if (c->done) return;
  if (!bfs_open_level(a, p.level, p.slot)) return;  // (an empty frontier: done = 1, levels = level)
  c->slots += 1;
  c->flush_count[(p.slot + 1) & 1] = 0;
  bfs_slot_marks_clear(a, p.slot + 1);
  c->slot_level[(p.slot + 1) & 3] = p.level + 1;
  c->skip_build[p.slot & 3] = 0;
  if (p.dense) c->dense_slots += 1;
  if (p.cold || p.colds) { c->cold_slot = p.slot; c->cold_slots += 1; }
  if (p.vshort) c->vshort_slots += 1;
}

// Push of one slot, ONE launch: block 0 opens the level (or runs the chain of small levels and everybody else
// returns); the first `nstream` workgroups take the long rows -- from the unit blocks or by walking the long-row queue
// -- the others the short-row queue.  One launch per level instead of two (~6 us of device time each, measured), and
// the short rows start while the last long-row slices drain.  Profiling runs (bfs_fused_state_t::time_kernels) launch
// the parts separately so that each can be bracketed by events.
// COLDT: probe the bitmap word of neighbours outside the LDS prefix (big graphs: many cold endpoints) or mark them
// untested (k_bfs_build tests the bitmap anyway).  PART: 0 all, 1 opener / chain only, 2 long rows only, 3 short rows only.
// (The 18 spilled SGPRs of <false, 0> are the chain body of block 0, inlined.  Calling it instead -- __attribute__((noinline)), round 5
//  -- takes the SGPR spills to 0 and brings 72 spilled VGPRs and 1 560 bytes of scratch: a call makes the kernel provide for the
//  callee's registers under its own 64-VGPR budget.  SGPR spills go to VGPR lanes, not to memory; left as it is.)
template <bool COLDT, int PART>
__global__ __launch_bounds__(1024, 8) void k_bfs_push(bfs_fused_args_t a, int arg, u32 nstream) {
  const bfs_slot_plan_t p = bfs_slot_plan(a, arg);
  if (p.chained) {
    if ((PART == 0 || PART == 1) && blockIdx.x == 0) bfs_chain_body<1024>(a, p.slot, p.level);
    return;
  }
  if ((PART == 0 || PART == 1) && blockIdx.x == 0 && threadIdx.x == 0) bfs_slot_open(a, p);
  if (p.empty || PART == 1) return;
  if (p.pulls) {
    // a bottom-up level of a direction-optimising run: the whole grid sweeps the vertices (bfs_fused_pull.hpp) -- inside this
    // launch unless the host asked for the sweep as a launch of its own (a.merged_pull == 0)
    if (a.merged_pull && PART != 3) {
      extern __shared__ __attribute__((aligned(16))) char smem_pull[];
      bfs_pull_body<1024>(a, blockIdx.x, gridDim.x, (unsigned long long*)smem_pull);
    }
    return;
  }
  // Which part this workgroup takes: (graphs with cold-edge lists) the workgroups of the cold pass, then nstream
  // workgroups for the long rows, the others the short rows.
  const u32 ncold = (!COLDT && a.cold_dst && (PART == 0 || PART == 2)) ? a.cold_wgs[a.cold_slices] : 0u;   // (PART 2: with the long rows)
  // grid: [cold pass][long rows][short rows] -- the cold workgroups first: they are few and short, and the launch does
  // not end on them
  if (blockIdx.x < ncold) {
    if (p.cold || p.colds) bfs_cold_body<1024>(a, p.slot, blockIdx.x, p.level, p.cold, p.colds);
    return;
  }
  if (a.cold_all && p.cold) return;               // (a flat graph's sweep level: the lists hold every entry)
  const u32 blk = blockIdx.x - ncold, nblk = gridDim.x - ncold;
  // (two other ways to deal the two halves were lab shapes until round 5 and lost in rounds 3 and 4: long- and short-row
  //  workgroups alternating, 0.52 against 0.41 ms per traversal -- the two bodies side by side on a CU are slower than one after
  //  the other --, and both dense paths in the SAME workgroups, 0.4055 / 0.4012 and again 0.3582 / 0.3481; HISTORY.md 3.1)
  const bool long_part = PART == 2 || (PART == 0 && blk < nstream);
  if (long_part) {
    const u32 bi = blk;
    if (!COLDT && p.dense) bfs_dense_body<1024, BFS_DENSE_HOTW>(a, p.slot, bi, nstream, p.level, p.cold);
    else bfs_stream_body<1024, BFS_STREAM_HOTW2, 8, COLDT, false, true>(a, p.slot, bi, nstream, p.level);
  } else {
    const u32 first = PART == 0 ? nstream : 0u;
    const u32 bi = blk - first;
    const u32 nb = nblk - first;
    if (!COLDT && p.vshort) bfs_vshort_body<1024, BFS_DENSE_HOTW>(a, p.slot, bi, nb, p.level, p.colds);
    else bfs_wave_body<1024, BFS_WAVE_HOTW, COLDT, false>(a, p.slot, bi, nb, p.level);
  }
}

// The chain of small levels as a launch of its own, ONE workgroup with (nearly) all of a CU's LDS, IN FRONT of a slot:
//   * at the start of a traversal (slot 0, behind the init kernel): the source's level and whatever small levels follow
//     it cost a 1-workgroup launch instead of a push launch over the whole grid plus a queue build that finds nothing to
//     do (~19 us -> ~10 on RMAT-22);
//   * at its end (slots >= tail_from, which the host learns from the previous traversal of the graph, and once more
//     behind the last slot of a batch): the stragglers -- RMAT-22: 10 000 edges, then 1 300 -- used to take two slots,
//     their own push launches over 1024 workgroups and sweeps of the queue build.  With twice the list capacity of the
//     chain inside a push launch (BFS_CHAIN_CAP_BIG: the LDS is this workgroup's alone) the levels of up to
//     chain_big_edges edges run here, back to back, and the traversal ends without another slot.
// The first level that is not small is left in the SAME slot's queues (bfs_chain_body<., true, .>): the slot's push
// launch opens it.  Nothing to do (a big level, an empty or lazy slot, the traversal over): returns at once (~2.5 us).
__global__ __launch_bounds__(1024) void k_bfs_chain_inplace(bfs_fused_args_t a, int arg) {
  const bfs_ctrl_t* const c = a.ctrl;
  int slot, level;
  bfs_resolve(c, arg, slot, level);
  const u64 cur = c->cursor[slot % 3], lcur = c->lcursor[slot % 3];
  if (c->done || ((cur | lcur) >> BFS_VSHIFT) == 0) return;         // (an empty frontier: the slot's opener ends the traversal)
  if (c->lazy_slot == slot) return;                                  // (the build before wrote no queues: nothing to stage in)
  if (!bfs_level_is_chained(a, cur, lcur, c->ledges[slot % 3], a.chain_big_edges, BFS_CHAIN_CAP_BIG)) return;
  bfs_chain_body<1024, true, BFS_CHAIN_CAP_BIG>(a, slot, level);
}

#ifdef MGX_LAB
// The instrumented stream kernel (MGX_BFS_FLAGS: parts of the body switched off for measurements; results are wrong
// by design).  Queue walk only.
__global__ __launch_bounds__(1024, 8) void k_bfs_push_stream_diag(bfs_fused_args_t a, int arg) {
  const bfs_slot_plan_t p = bfs_slot_plan(a, arg);
  if (p.empty || p.chained) return;
  bfs_stream_body<1024, BFS_STREAM_HOTW2, 8, false, true, true>(a, p.slot, blockIdx.x, gridDim.x, p.level);
}
#endif  // MGX_LAB

// Explicit-level variant for the partitioned path (bfs_dist2.hpp): slot == level, the level's bookkeeping rides on
// the launch (open_here == 2: a rank of a partitioned run), no chain; unit blocks when the rank built them.
template <bool COLDT>
__global__ __launch_bounds__(1024, 8) void k_bfs_push_level(bfs_fused_args_t a, int level, u32 nstream, int open_here, u32 ncold) {
  // a level with no more edges than the rank's id list has room for writes its discoveries there itself (bfs_fused_sparse.hpp):
  // no marks for k_d2_newbits to sweep.  Grid-uniform: read from the ring entry of the level BEFORE the opener touches anything
  // (it clears the entry two levels ahead only).
  // (measurements, MGX_DIST_PUSH_SPLIT=1: the three parts of the grid as three launches -- bits 4 / 5 / 6 of open_here switch the
  //  cold pass / the long rows / the short rows of THIS launch off; the level's bookkeeping rides on the first)
  if (bfs_d2_frozen(a.ctrl, level)) return;        // (a speculative plan ran past a level that has to wait for the host: bfs_dist2.hpp)
  const int skip = open_here >> 4;
  open_here &= 15;
  const bool appends = bfs_d2_level_appends(a, level);
  if (open_here && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
    bfs_begin_level(a, level, open_here == 2);
    bfs_slot_marks_clear(a, level + 1);
    a.ctrl->flush_count[(level + 1) & 1] = 0;
    a.ctrl->d2_append_level = appends ? level : -1;
  }
  if (appends) {
    if (!(skip & 1)) bfs_d2_sparse_body<1024>(a, level, blockIdx.x, gridDim.x);     // (a split push: with its first launch)
    return;
  }
  // a rank that carries unit blocks of its rows (bfs_dist2.hpp: owners in GLOBAL ids, frontier_bits = the level's merged
  // discoveries of all ranks) reads a level that holds a large share of its long rows from them -- the queue-less body of
  // the single-GPU path; with cold-edge lists the first ncold workgroups take the entries behind the LDS prefix by slice
  // (bfs_fused_cold.hpp) and the unit-block body skips them.  Grid-uniform: the long-row cursor of the level is stable
  // while the level runs.
  // (a graph big enough for the cold TEST -- most endpoints behind the prefix -- reads the unit blocks only together with the
  //  cold-edge lists: marking those entries untested is what the test is there to avoid)
  const bool dense = bfs_long_is_dense(a, a.ctrl, level, a.ctrl->lcursor[level % 3]) && (!COLDT || a.cold_dst != nullptr);
  const bool cold = dense && a.cold_dst != nullptr;
  if (blockIdx.x < ncold) {
    if (cold && !(skip & 1)) {
      if (blockIdx.x == 0 && threadIdx.x == 0) { a.ctrl->cold_slot = level; a.ctrl->cold_slots += 1; }   // (k_d2_newbits: OR the slices' flush bitmaps in)
      bfs_cold_body<1024>(a, level, blockIdx.x, level, true, false);
    }
    return;
  }
  const u32 blk = blockIdx.x - ncold, nblk = gridDim.x - ncold;
  if (blk < nstream ? (skip & 2) != 0 : (skip & 4) != 0) return;
  if (blk < nstream) {
    if (dense) {
      if (blk == 0 && threadIdx.x == 0) a.ctrl->dense_slots += 1;
      bfs_dense_body<1024, BFS_DENSE_HOTW>(a, level, blk, nstream, level, cold);
    } else bfs_stream_body<1024, BFS_STREAM_HOTW2, 8, COLDT, false, true>(a, level, blk, nstream, level);
  } else if (a.d2_front && bfs_short_is_dense(a, a.ctrl, level, a.ctrl->cursor[level % 3])) {
    // (grid-uniform) a level that holds a large share of the rank's short rows walks them vertex by vertex, by the frontier over
    // its LOCAL rows the merge left behind (bfs_fused_vshort.hpp; the queue is not looked at)
    if (blk == nstream && threadIdx.x == 0) a.ctrl->vshort_slots += 1;
    bfs_vshort_body<1024, BFS_DENSE_HOTW, COLDT>(a, level, blk - nstream, nblk - nstream, level);
  } else bfs_wave_body<1024, BFS_WAVE_HOTW, COLDT, false>(a, level, blk - nstream, nblk - nstream, level);
}

inline void bfs_set_kernel_attributes() {
  static unsigned char seen[64] = {};
  device_once_t once(seen);
  if (!once) return;
#define MGX_SET_LDS(K_) MGX_HIP(hipFuncSetAttribute((const void*)K_, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
  MGX_SET_LDS((k_bfs_push<false, 0>)); MGX_SET_LDS((k_bfs_push<true, 0>));
  MGX_SET_LDS((k_bfs_push<false, 1>)); MGX_SET_LDS((k_bfs_push<true, 1>));
  MGX_SET_LDS((k_bfs_push<false, 2>)); MGX_SET_LDS((k_bfs_push<true, 2>));
  MGX_SET_LDS((k_bfs_push<false, 3>)); MGX_SET_LDS((k_bfs_push<true, 3>));
#ifdef MGX_LAB
  MGX_SET_LDS(k_bfs_push_stream_diag);
#endif
  MGX_SET_LDS(k_bfs_chain_inplace);
  // (k_bfs_mini has static LDS too: the attribute carries what it needs, not the whole 160 KB)
  MGX_HIP(hipFuncSetAttribute((const void*)k_bfs_mini<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bfs_mini_lds_bytes()));
  MGX_SET_LDS(k_bfs_push_level<false>);
  MGX_SET_LDS(k_bfs_push_level<true>);
#undef MGX_SET_LDS
}

// -1: decide by size (cold test when the bitmap is at least 8 x the LDS prefix), 0 / 1: forced (MGX_BFS_COLD_TEST)
inline bool bfs_cold_test(int n, int forced) {
  if (forced >= 0) return forced != 0;
  return (long long)n >= 8ll * 32 * BFS_STREAM_HOTW;
}

// Shapes that were measured and dropped, RMAT-22 (stream kernel of the big level / whole BFS at the time):
//   stream  2 workgroups x 1024 threads per CU = 32 waves, 80 KB of bitmap each, 8 non-temporal loads per lane (kept):
//           167 us / 0.66 ms; 16 loads per lane 190 / 0.68; 1 workgroup per CU with 160 KB of bitmap 213-217 / 0.69-0.71;
//           cached instead of non-temporal col_indices loads 0.582 vs 0.565 ms per traversal; 16-byte-per-lane reads
//           (sub-rounds of 256 consecutive edges) 190-245 us; on partitioned RMAT-25 (cold test) one workgroup per CU
//           with 160 KB was 3 % faster than two with 80 KB -- not enough for a second merged shape;
//   wave    2 x 1024 threads per CU (kept) 67 us, 2 x 512 threads with more bitmap in LDS 82 us; non-temporal loads: same;
//   build   512 threads (kept) 0.445 ms per traversal, 256: 0.453-0.461, 1024: 0.476.
template <int PART>
inline void bfs_launch_push_part(const bfs_fused_args_t& a, int arg, standard_context_t& ctx, bool coldt, unsigned grid, u32 nstream) {
  if (coldt) hipLaunchKernelGGL((k_bfs_push<true, PART>), dim3(grid), dim3(1024), bfs_push_lds_bytes(), ctx.stream(), a, arg, nstream);
  else hipLaunchKernelGGL((k_bfs_push<false, PART>), dim3(grid), dim3(1024), bfs_push_lds_bytes(), ctx.stream(), a, arg, nstream);
}

// explicit-level push of the partitioned path
inline void bfs_launch_push(const bfs_fused_args_t& a, int level, standard_context_t& ctx, int open_here, bool coldt, u32 grid_div = 1) {
  // (grid_div: a rank with a small shard takes fewer, fatter workgroups -- every one of them copies 80 KB of bitmap into LDS first)
  if (grid_div < 1u) grid_div = 1u;
  u32 per = (u32)ctx.num_cus * 2 / grid_div;
  if (per < 32u) per = 32u;
  const u32 nstream = a.long_min > 0 ? per : 0u;
  const u32 nwave = per;
  const u32 ncold = a.cold_dst ? a.cold_wgs[a.cold_slices] : 0u;
  if (coldt) hipLaunchKernelGGL(k_bfs_push_level<true>, dim3(ncold + nstream + nwave), dim3(1024), bfs_push_lds_bytes(), ctx.stream(), a, level, nstream, open_here, ncold);
  else hipLaunchKernelGGL(k_bfs_push_level<false>, dim3(ncold + nstream + nwave), dim3(1024), bfs_push_lds_bytes(), ctx.stream(), a, level, nstream, open_here, ncold);
}

// Everything a traversal's launches need that does not depend on the source: the kernel arguments (which bodies are
// available on this graph / layout, thresholds, buffers) and the grid shapes.  Made once per call (per batch of sources).
struct bfs_launch_plan_t {
  bfs_fused_args_t a;
  bool coldt = false, build2_ok = false;
  bool minis = false;       // M launches (bfs_fused_mini.hpp) in front of the first and behind the last device-wide slot of a traversal
  u32 nstream = 0, nwave = 0, ncold = 0;
  int lab_flags = 0;
  long long nwords = 0;
  int mode = 0;
  int* labels = nullptr;
};

inline bfs_launch_plan_t bfs_fused_plan(bfs_fused_state_t& st, const int* row_offsets, const int* col_indices, int* labels,
                                        standard_context_t& ctx, const bfs_layout_t* layout, int mode, float alpha,
                                        const int* in_offsets, const int* in_indices) {
  bfs_launch_plan_t plan;
  bfs_fused_args_t& a = plan.a;
  const bfs_run_opts_t& opt = st.opts;        // (environment switches: read once, when the handle's state was made)
  const bool relabelled = layout && layout->row_offsets;
  a.row_offsets = (const u32*)(relabelled ? layout->row_offsets : row_offsets);
  a.col_indices = relabelled ? layout->col_indices : col_indices;
  a.old_of_new = relabelled ? layout->old_of_new : nullptr;
  a.new_of_old = relabelled ? layout->new_of_old : nullptr;
  bfs_set_kernel_attributes();
  a.labels = labels;
  a.visited = st.visited.data();
  a.mark = st.mark.data();
  a.frontier_bits = st.frontier_bits.data();
  a.mode = mode;
  a.alpha = alpha;
  a.in_offsets = (const u32*)(relabelled ? layout->row_offsets : (in_offsets ? in_offsets : row_offsets));
  a.in_indices = relabelled ? layout->col_indices : (in_indices ? in_indices : col_indices);
  for (int i = 0; i < 2; ++i) {
    a.fr_row[i] = st.fr_row[i].data(); a.fr_off[i] = st.fr_off[i].data();
    a.lq_row[i] = st.lq_row[i].data(); a.lq_off[i] = st.lq_off[i].data();
  }
  a.long_min = st.long_min;
  a.hot_min_edges = st.hot_min_edges;
  a.ctrl = st.ctrl.data();
  a.n = st.n;
  const int lab_flags = MGX_LAB_GET(opt, flags, 0);      // (lab builds: the instrumented stream kernel)
#ifdef MGX_LAB
  a.flags = lab_flags;
#endif
  a.count_marks = (st.count_marks || st.time_kernels == 1) ? 1 : 0;
  // unit blocks: only for the CSR they were built from, with the long-row threshold they were built for
  // (the entries at 32 bits, or their 24-bit copy when the run may read it: a layout that carries the copy has dropped the former)
  const bool units_avail = relabelled && (layout->ub_col || (layout->ub_col24 && st.opts.pack24)) && layout->ub_owner && layout->ub_units > 0 &&
                           layout->ub_min_degree == st.long_min && st.long_min > 0 && !lab_flags;
  // Probing the bitmap word of cold neighbours (instead of marking them untested) pays on big graphs WITHOUT the unit
  // blocks -- the partitioned ranks, a caller-made layout.  With them the unit-block body, the cold-edge pass and the
  // lazy builds win at every size measured: RMAT-23 391 against 272 GTEPS, RMAT-24 386 / 253, RMAT-25 187 / 179.
  // A FLAT graph (round 5: a uniform random graph of 2^22 vertices, every row ~32 entries): the LDS prefix holds 652 K of its
  // vertices and nearly every endpoint lies behind it -- the layout built no cold-edge lists (they would be most of the graph), so
  // the unit-block body would mark six entries in seven untested, a byte store each.  Probing the bitmap (512 KB: it lives in the
  // L2s) wins there: 1.35 against 2.02 ms per traversal.
  // (the layout says so: it counted the long rows' cold entries and found more than a quarter of them cold.  NOT "no lists": RMAT-20
  //  has none either -- a handful of cold entries -- and lost 19 % to this rule while it read "no lists": 0.1915 -> 0.2272 ms)
  const bool flat = units_avail && layout->cold_majority;
  const bool coldt = opt.cold_test >= 0 ? opt.cold_test != 0 : ((bfs_cold_test(a.n, -1) && !units_avail) || flat);
  const bool units = units_avail && !coldt;
  a.ub_col = units ? layout->ub_col : nullptr;
  a.ub_col24 = (units && st.opts.pack24) ? layout->ub_col24 : nullptr;
  a.ub_owner = units ? layout->ub_owner : nullptr;
  a.ub_units = units ? (u32)layout->ub_units : 0u;
  a.ub_units_pad = units ? (u32)layout->ub_units_pad : 0u;
  // (with the 24-bit copy a unit costs three quarters of the bytes: the unit-block body wins from an eighth of the units on
  //  -- RMAT-22, per call: 1/2 0.3282, 1/4 0.3262, 1/8 0.3215, 1/16 0.3250 ms; with 32-bit entries 1/2 was best)
  // k_bfs_build2 reads a thread's 17 row offsets and 16 layout ids with 16-byte loads: borrowed arrays must be aligned
  const bool build2_ok = ((uintptr_t)a.row_offsets % 16 == 0) && ((uintptr_t)a.old_of_new % 16 == 0);
  // cold-edge lists (bfs_fused_cold.hpp): with the unit blocks they were cut from, the prefix they were cut behind, and a
  // queue build that knows their bitmaps
  // (the pairs at 8 bytes, or every slice packed at 4 and a run that may read those)
  const bool cold_all_packed = layout && layout->cold_pk && layout->cold_cbase && layout->cold_slices > 0 &&
                               (layout->cold_pk_mask & (layout->cold_slices >= 64 ? ~0ull : ((1ull << layout->cold_slices) - 1ull))) ==
                                   (layout->cold_slices >= 64 ? ~0ull : ((1ull << layout->cold_slices) - 1ull));
  const bool cold_pairs_ok = layout && ((layout->cold_pairs8 && layout->cold_dst) || (cold_all_packed && opt.cold_pack));
  const bool flat_lists = layout && layout->cold_all && layout->cold_hot_n == 0u && cold_all_packed && opt.cold_pack && layout->cold_pairs_total > 0;
  const bool cold_lists = units && cold_pairs_ok && layout->cold_slices > 0 &&
                          (layout->cold_hot_n == (unsigned)(BFS_DENSE_HOTW * 32) || flat_lists) && (!layout->cold_all || flat_lists) &&
                          layout->cold_long_min == st.long_min && build2_ok && !opt.build_list && opt.cold != 0 &&
                          !MGX_LAB_GET(opt, dense_diag, 0);
  // ... and then the unit blocks WITHOUT the lists' entries (every level that reads unit blocks runs the cold-edge pass: the body
  // reads those entries only to skip them), at 24 bits whatever the graph's size.  Round 5, as the partitioned ranks have had
  // them since round 4 (mgx_capi.hip): see the note at the layout's builder for what they are worth.
  const bool hot_units = cold_lists && st.opts.pack24 && st.opts.hot_units && layout->ubh_col24 && layout->ubh_owner && layout->ubh_units > 0 &&
                         (opt.dense < 0 || opt.dense > 0);
  if (hot_units) {
    a.ub_col24 = layout->ubh_col24; a.ub_owner = layout->ubh_owner;     // (a.ub_col: the full blocks' -- not read by the 24-bit body)
    a.ub_units = (u32)layout->ubh_units; a.ub_units_pad = (u32)layout->ubh_units_pad;
  }
  a.ub_hot_only = hot_units ? 1u : 0u;
  a.ubf_col24 = hot_units ? layout->ub_col24 : nullptr; a.ubf_owner = hot_units ? layout->ub_owner : nullptr;
  a.ubf_units_pad = hot_units ? (u32)layout->ub_units_pad : 0u;
  const bool packed = units && st.opts.pack24 && (layout->ub_col24 != nullptr || hot_units);
  a.dense_div = !units ? 0u : (opt.dense >= 0 ? (u32)opt.dense : (packed ? 8u : st.dense_div));
  // short rows vertex by vertex: the layout's own degree-sorted CSR with its padding, the threshold it was cut for
  const bool vs = relabelled && layout->vs_dummy != 0 && layout->vs_long_min == st.long_min && st.long_min > 0 && !coldt && !lab_flags &&
                  layout->vs_edges > 0;
  for (int i = 0; i < 4; ++i) a.vs_v[i] = vs ? layout->vs_v[i] : 0u;
  a.vs_v9 = (vs && layout->vs_v9 >= layout->vs_v[1] && layout->vs_v9 <= layout->vs_v[2]) ? layout->vs_v9 : (vs ? layout->vs_v[2] : 0u);   // (unknown: nobody in the two-lane class)
  a.vs_edges = vs ? layout->vs_edges : 0u;
  a.vs_dummy = vs ? layout->vs_dummy : 0u;
  a.vs_div = !vs ? 0u : (opt.vshort >= 0 ? (u32)opt.vshort : st.vshort_div);
#ifdef MGX_LAB
  a.dense_diag = opt.dense_diag;
  a.build_diag = opt.build_diag;
#endif
  if (!st.slot_marks.size()) st.slot_marks = mem_t<u32>((size_t)2 * BFS_MARK_CTRS * BFS_MARK_STRIDE, ctx);
  a.slot_marks = st.slot_marks.data();
  a.merged_pull = (mode == 1 && opt.merged_pull && opt.merged && !lab_flags) ? 1 : 0;
  // deferred hot marks (bfs_hot_epilogue): the flush buffers are allocated at the first traversal that may use them
  const long long defer = opt.defer >= 0 ? opt.defer : (long long)st.defer_min_marks;
  if (defer > 0 && !lab_flags && !st.flush_buf.size()) st.flush_buf = mem_t<u32>((size_t)BFS_FLUSH_MAX * BFS_FLUSH_WORDS, ctx);
  a.flush_buf = (defer > 0 && !lab_flags) ? st.flush_buf.data() : nullptr;
  a.defer_min_marks = (u32)defer;
  {
    // the deferred range: whole runs of 1024 vertices (the queue build ORs the flush buffers run by run), at most the buffers' stride
    int dw = opt.defer_words >= 0 ? opt.defer_words : BFS_FLUSH_WORDS;
    if (dw > BFS_FLUSH_WORDS) dw = BFS_FLUSH_WORDS;
    a.defer_words = (u32)(dw / 32 * 32);
  }
  a.defer_reach_mul = (u32)opt.defer_mul; a.defer_reach_div = (u32)opt.defer_div;
  a.chain_max_edges = (mode != 0 && !opt.do_chain) ? 0u : (opt.chain >= 0 ? (u32)(opt.chain > BFS_CHAIN_CAP ? BFS_CHAIN_CAP : opt.chain) : st.chain_max_edges);
  const long long nwords = ((long long)st.n + 31) / 32;
  // in-place chain launches (k_bfs_chain_inplace): in front of slot 0, of the slots from tail_from on, behind a batch
  a.chain_big_edges = (a.chain_max_edges && opt.seed_chain) ? (opt.chain_big >= 0 ? (u32)(opt.chain_big > BFS_CHAIN_CAP_BIG ? BFS_CHAIN_CAP_BIG : opt.chain_big) : st.chain_big_edges) : 0u;
  const bool cold = cold_lists && a.dense_div;
  // (cold_dst != NULL is what "this run has cold-edge lists" reads as everywhere: a layout without the 8-byte pairs -- every slice
  //  packed, bfs_cold_body never dereferences them -- hands the packed words' address instead)
  a.cold_owner = cold ? (layout->cold_pairs8 ? layout->cold_owner : (const int*)layout->cold_pk) : nullptr;
  a.cold_dst = cold ? (layout->cold_pairs8 ? layout->cold_dst : (const int*)layout->cold_pk) : nullptr;
  a.cold_slices = cold ? layout->cold_slices : 0;
  a.cold_all = (cold && flat_lists) ? 1u : 0u;
  a.cold_all_pairs = a.cold_all ? layout->cold_pairs_total : 0ull;
  const bool cold_pk = cold && layout->cold_pk && layout->cold_cbase && opt.cold_pack;
  a.cold_pk = cold_pk ? layout->cold_pk : nullptr; a.cold_cbase = cold_pk ? layout->cold_cbase : nullptr;
  a.cold_pk_mask = cold_pk ? layout->cold_pk_mask : 0ull; a.cold_ranks = 1;
  for (int i = 0; i <= BFS_COLD_MAX_SLICES; ++i) a.cold_cb[i] = cold_pk ? layout->cold_cb[i] : 0u;
  for (int i = 0; i < BFS_COLD_MAX_SLICES; ++i) a.cold_lo[i] = cold ? layout->cold_lo[i] : 0u;
  for (int i = 0; i <= BFS_COLD_MAX_SLICES; ++i) { a.cold_off[i] = cold ? layout->cold_off[i] : 0u; a.cold_wgs[i] = cold ? layout->cold_wgs[i] : 0u; }
  // ... and of the short rows, for the levels that walk them vertex by vertex
  const bool colds = cold && a.vs_div && layout->colds_dst && opt.cold != 2;
  a.colds_owner = colds ? layout->colds_owner : nullptr;
  a.colds_dst = colds ? layout->colds_dst : nullptr;
  for (int i = 0; i <= BFS_COLD_MAX_SLICES; ++i) a.colds_off[i] = colds ? layout->colds_off[i] : 0u;
  const size_t cold_words = cold ? (size_t)layout->cold_wgs[layout->cold_slices] * BFS_COLD_WORDS : 0;
  if (cold && st.cold_flush.size() < cold_words) { ctx.synchronize(); st.cold_flush = mem_t<u32>(cold_words, ctx); }
  a.cold_flush = cold ? st.cold_flush.data() : nullptr;
  // lazy queues (bfs_build_is_lazy): only k_bfs_build2 knows them, and only when both queue-less bodies are available
  a.lazy_pull = (mode == 1 && build2_ok && !opt.build_list && opt.lazy != 0) ? 1 : 0;
  // (direction-optimising runs too: a level behind a lazy build either pulls -- no queue needed -- or takes the queue-less bodies)
  a.lazy_div = (a.dense_div && a.vs_div && build2_ok && !opt.build_list && !a.cold_all) ? (opt.lazy >= 0 ? (u32)opt.lazy : st.lazy_div) : 0u;      // (a flat graph's levels keep their queues: what is not swept is walked)
  plan.coldt = coldt;
  plan.build2_ok = build2_ok;
  // M launches from RMAT-22's size on: what one costs does not depend on the graph, what the device-wide slot it replaces costs
  // does (its queue build sweeps all n marks).  Measured, batches of 32 sources: RMAT-23 0.703 against 0.720 ms per traversal
  // without them, RMAT-22 equal to 1 % better; with them at every size RMAT-21 / 18 were 1.4 / 2.4 % worse and RMAT-20 0.236
  // against 0.195 ms (levels of 30-130 K edges are a large share of such a graph; two of 16 sources re-run).  MGX_BFS_MINI=2
  // forces them (the tests' graphs are small)
  plan.minis = mode == 0 && opt.mini != 0 && a.chain_big_edges != 0u && opt.merged && !lab_flags && (st.n >= (1 << 22) || opt.mini == 2);
  plan.nstream = a.long_min > 0 ? (u32)ctx.num_cus * 2 : 0u;
  plan.nwave = (u32)ctx.num_cus * 2;
  plan.ncold = (!coldt && a.cold_dst) ? a.cold_wgs[a.cold_slices] : 0u;
  plan.lab_flags = lab_flags;
  plan.nwords = nwords;
  plan.mode = mode;
  plan.labels = labels;
  return plan;
}

// does a run on this handle look at the sources' shapes at all?  (the callers resolve them only then: a launch and a wait per batch of
// sources the cache has not seen)
inline bool bfs_wants_src_shapes(const bfs_fused_state_t& st, int mode) {
  return mode == 0 && st.opts.src_plan && st.opts.tail_chain && st.opts.mini != 0 && st.opts.merged && st.opts.seed_chain &&
         (st.n >= (1 << 22) || st.opts.mini == 2);
}
// idx: the source's position in the call's source list (the row of layout->src_shapes that is its)
inline int bfs_classify_source(const bfs_fused_state_t& st, const bfs_launch_plan_t& plan, const bfs_layout_t* layout, int src, int idx = 0) {
  if (!layout || !layout->src_shapes || !plan.minis || plan.mode != 0 || !st.opts.src_plan || !st.opts.tail_chain) return BFS_SRC_UNKNOWN;
  const bfs_fused_args_t& a = plan.a;
  if (layout->src_shapes_long_min != a.long_min || src < 0 || src >= a.n) return BFS_SRC_UNKNOWN;
  const unsigned* const sh = layout->src_shapes + (size_t)idx * 4;
  const u64 deg = sh[0], e1 = sh[1], s1 = sh[2], l1 = sh[3];
  if (deg == 0 || e1 == 0xFFFFFFFFull) return BFS_SRC_UNKNOWN;
  // both levels run before a quarter of the vertices is reached (the chain's and the M launch's EARLY limits apply)
  if ((1ull + s1 + l1) * 4ull >= (u64)(u32)a.n) return BFS_SRC_UNKNOWN;
  const u64 lim = a.chain_big_edges < BFS_CHAIN_EARLY_EDGES ? a.chain_big_edges : BFS_CHAIN_EARLY_EDGES;
  const u64 chain_cap = lim < (u64)BFS_CHAIN_CAP_BIG ? lim : (u64)BFS_CHAIN_CAP_BIG;
  u64 E, rs, rl;
  if (deg > chain_cap) {                       // the source's own row is more than the in-place chain takes
    const bool is_long = a.long_min > 0 && deg >= (u64)a.long_min;
    E = deg; rs = is_long ? 0 : 1; rl = is_long ? 1 : 0;
  } else if (e1 > chain_cap) {                 // level 0 is chained, level 1 is what the chain leaves
    E = e1; rs = s1; rl = l1;
  } else {
    return BFS_SRC_UNKNOWN;
  }
  const bool mini = rl <= (u64)BFS_MINI_LCAP && rs <= (u64)BFS_MINI_SHORT_ROWS && E <= (u64)BFS_MINI_EDGES_EARLY;
  return mini ? BFS_SRC_ABSORB : BFS_SRC_SKIP;
}
// device-wide slots a traversal of class `cls` gets: what the last traversals of the class needed, else the graph's hint
inline int bfs_class_slots(const bfs_fused_state_t& st, int cls) {
  if (cls == BFS_SRC_UNKNOWN) return st.slots_hint;
  if (st.cls_at[cls] > 0) return st.cls_max[cls];
  // no traversal of this class yet: a SKIP source needs the slot an ABSORB source's M launch saves; an ABSORB source
  // without history gets the graph's (conservative) number
  if (cls == BFS_SRC_SKIP && st.cls_at[BFS_SRC_ABSORB] > 0) return st.cls_max[BFS_SRC_ABSORB] + 1 > st.slots_hint ? st.cls_max[BFS_SRC_ABSORB] + 1 : st.slots_hint;
  return st.slots_hint;
}

// the launches of a traversal on the product path (no events, merged push): init + the chain of the small levels at the
// start; one slot; the chain behind the last slot of a batch.  prev_head: see k_bfs_fused_init.
inline void bfs_enqueue_chain_inplace(const bfs_launch_plan_t& plan, int slot, hipStream_t s) {
  hipLaunchKernelGGL(k_bfs_chain_inplace, dim3(1), dim3(1024), bfs_chain_lds_bytes(BFS_CHAIN_CAP_BIG), s, plan.a, bfs_slot_arg(slot));
}
inline void bfs_enqueue_mini(const bfs_launch_plan_t& plan, int slot, hipStream_t s) {
  hipLaunchKernelGGL(k_bfs_mini<1024>, dim3(BFS_MINI_WGS), dim3(1024), bfs_mini_lds_bytes(), s, plan.a, bfs_slot_arg(slot));
}
// init, the chain of the tiny levels at the start, and (M launches on) the first mid-size level: returns the first slot
// that is still to be launched
inline int bfs_enqueue_start(const bfs_fused_state_t& st, const bfs_launch_plan_t& plan, int src, standard_context_t& ctx,
                             const bfs_ctrl_t* prev_ctrl = nullptr, bfs_ctrl_t* prev_head = nullptr, int head_words = 0,
                             bool front_mini = true) {
  hipStream_t s = ctx.stream();
  hipLaunchKernelGGL(k_bfs_fused_init, dim3(grid_for(((long long)st.n + 3) / 4, BLOCK, ctx.num_cus * 8)), dim3(BLOCK), 0, s, plan.a, src,
                     plan.nwords, prev_ctrl, prev_head, head_words);
  if (plan.a.chain_big_edges) bfs_enqueue_chain_inplace(plan, 0, s);
  if (plan.minis && front_mini) { bfs_enqueue_mini(plan, 0, s); return 1; }
  return 0;
}
inline void bfs_enqueue_build(const bfs_fused_state_t& st, const bfs_launch_plan_t& plan, int arg, hipStream_t s) {
  if (st.opts.build_list || !plan.build2_ok)
    hipLaunchKernelGGL((k_bfs_build<512, true>), dim3(bfs_build_grid(st.n, 512)), dim3(512), 0, s, plan.a, arg,
                       (const u32*)nullptr, plan.labels, st.n, 1, 0, 1);
  else
    hipLaunchKernelGGL(k_bfs_build2<512>, dim3(bfs_build_grid(st.n, 512)), dim3(512), 0, s, plan.a, arg, plan.labels, st.n);   // (2 workgroups per CU overlap their phases)
}
inline void bfs_enqueue_slot(const bfs_fused_state_t& st, const bfs_launch_plan_t& plan, int slot, standard_context_t& ctx) {
  hipStream_t s = ctx.stream();
  const int arg = bfs_slot_arg(slot);
  if (plan.a.chain_big_edges && slot > 0 && slot >= st.tail_from && st.opts.tail_chain && st.opts.tail_front && !plan.minis)
    bfs_enqueue_chain_inplace(plan, slot, s);
  bfs_launch_push_part<0>(plan.a, arg, ctx, plan.coldt, plan.nstream + plan.ncold + plan.nwave, plan.nstream);
  if (plan.mode == 1 && !plan.a.merged_pull)
    hipLaunchKernelGGL(k_bfs_pull_level<256>, dim3(ctx.num_cus * 8), dim3(256), 0, s, plan.a, arg);
  bfs_enqueue_build(st, plan, arg, s);
}

// What the NEXT traversal of this graph should enqueue, from the level sizes of the one whose control block `hc` holds:
// a slot for every level that is too big for the in-place chain (sources differ, the level structure of a graph hardly
// does; a slot too many is two idle launches, ~9 us; one too few a second batch, ~25 us) -- the most any of the last four
// traversals needed.  In-place chain launches from the first slot behind them on, i.e. behind the batch: a chain launch
// that finds a big level costs 5 us (its reads of the control block miss behind the build's atomics), and on RMAT-22 the
// level behind the last big one is small for a minority of the sources only.
inline void bfs_learn_slots(bfs_fused_state_t& st, const bfs_fused_args_t& a, const bfs_ctrl_t* hc, int mode, int trace_avail, bool minis = false,
                            int cls = BFS_SRC_UNKNOWN) {
  st.slots_hint = hc->slots > 0 ? hc->slots : 1;    // slots that found work
  st.tail_from = 1 << 30;
  if (!(a.chain_big_edges && st.opts.tail_chain)) return;
  const int lv = hc->levels < trace_avail ? hc->levels : trace_avail;     // (levels behind the part of the trace the host holds count as small)
  // 0 small (a chain launch runs it), 1 mid-size (an M launch, bfs_fused_mini.hpp), 2 a device-wide slot
  int kind[64];
  const int L = lv < 64 ? lv : 64;
  bool pulled = false;
  u64 reached = 1;                         // (vertices with edges reached before level l runs: what bfs_chain_edge_limit looks at, nearly)
  int extra_big = lv > 64 ? lv - 64 : 0;   // (deep traversals: their tails are chains of small levels)
  (void)extra_big;
  for (int l = 0; l < L; ++l) {
    const u64 t = hc->trace[l];
    if (l > 0) reached += t >> BFS_VSHIFT;
    const bool late = reached * 4ull >= (u64)(u32)a.n;
    const u64 lim = late || a.chain_big_edges < BFS_CHAIN_EARLY_EDGES ? a.chain_big_edges : BFS_CHAIN_EARLY_EDGES;
    bool small = (t >> BFS_VSHIFT) <= (u64)BFS_CHAIN_CAP_BIG && (t & BFS_EMASK) <= lim;
    // (the trace holds the level's vertices in all; the device rule looks at its long and its short rows separately)
    bool mid = (t >> BFS_VSHIFT) <= (u64)BFS_MINI_SHORT_ROWS && (t & BFS_EMASK) <= (u64)(late ? BFS_MINI_EDGES_LATE : BFS_MINI_EDGES_EARLY);
    if (mode == 1) {                       // direction-optimising: bottom-up levels (and everything behind the first) are device-wide
      const float unvisited = (float)((long long)a.n - (long long)reached);
      if (unvisited < (float)(long long)(t >> BFS_VSHIFT) * a.alpha) pulled = true;
      if (pulled) small = false;
      mid = false;
    }
    kind[l] = small ? 0 : (minis && mid ? 1 : 2);
  }
  // the launch sequence [chain][M] slots [M][chain]: what the two ends absorb needs no slot
  auto slots_needed = [&](bool front_mini) {
    int lo = 0, hi = L;
    while (lo < hi && kind[lo] == 0) ++lo;                    // the chain at the start
    if (minis && front_mini && lo < hi && kind[lo] == 1) ++lo;   // the M launch behind it
    while (hi > lo && kind[hi - 1] == 0) --hi;                // the chain behind the batch
    if (minis && hi > lo && kind[hi - 1] == 1) {              // the M launch in front of it ...
      --hi;
      while (hi > lo && kind[hi - 1] == 0) --hi;              // (... and small levels in front of THAT ride in a slot's push launch)
    }
    int k = 0;
    for (int l = lo; l < hi; ++l) if (kind[l] != 0) ++k;
    return k > 0 ? k : 1;
  };
  if (cls != BFS_SRC_UNKNOWN) {                               // what this class needs under ITS sequence (SKIP: no M launch in front)
    const int need_cls = slots_needed(cls != BFS_SRC_SKIP);
    if (need_cls > st.cls_max[cls]) st.cls_max[cls] = need_cls;
    st.cls_at[cls] += 1;
  }
  int need = slots_needed(true);                              // ... and the graph-wide sequence, whoever ran
  // (a traversal deeper than the trace the host holds -- 64 levels: a grid, a road network -- needed what the device counted: its
  //  first levels, all small, say nothing about the thousands behind them)
  if (hc->levels > L && hc->slots > need) need = hc->slots;
  st.recent_need[st.recent_at & 3] = need;
  st.recent_at += 1;
  int hint = 1;
  for (int i = 0; i < 4 && i < st.recent_at; ++i)
    if (st.recent_need[i] > hint) hint = st.recent_need[i];
  st.slots_hint = hint;
  st.tail_from = hint;
}

// Runs a whole BFS from `src` on the context's stream.  labels[] is (re)initialised here.  Returns with the
// stream synchronised and host_ctrl holding the final counters.
// mode/alpha: MGX_BFS_PUSH (0) or MGX_BFS_DIRECTION_OPT (1) with the reference's switch rule
// num_unvisited < frontier_length * alpha (bfs_enactor.hxx:68).  in_offsets/in_indices: in-edges for the
// bottom-up levels (pass the CSR for symmetric graphs, the reference's behaviour -- SURVEY F8).
inline void bfs_fused_run(bfs_fused_state_t& st, const int* row_offsets, const int* col_indices, int* labels,
                          int src, standard_context_t& ctx, const bfs_layout_t* layout = nullptr, int mode = 0,
                          float alpha = 0.f, const int* in_offsets = nullptr, const int* in_indices = nullptr) {
  hipStream_t s = ctx.stream();
  const bfs_run_opts_t& opt = st.opts;
  const bfs_launch_plan_t plan = bfs_fused_plan(st, row_offsets, col_indices, labels, ctx, layout, mode, alpha, in_offsets, in_indices);
  const bfs_fused_args_t& a = plan.a;
  const bool coldt = plan.coldt;
  const int lab_flags = plan.lab_flags;
  const int cls = bfs_classify_source(st, plan, layout, src);
  int slot = bfs_enqueue_start(st, plan, src, ctx, nullptr, nullptr, 0, cls != BFS_SRC_SKIP);          // (1 when an M launch took slot 0)
  auto chain_inplace = [&](int sl) { bfs_enqueue_chain_inplace(plan, sl, s); };
  st.level_kernel_ms = 0.0;
  st.level_kernel_launches = 0;
  st.wave_kernel_ms = 0.0;
  st.wave_kernel_launches = 0;
  st.stream_kernel_ms = 0.0;
  st.stream_kernel_launches = 0;
  st.batches = 0;
  const bool batch_events = st.time_kernels != 0 || st.time_batches;    // (an event costs ~6 us of stream gap)
  const u32 nstream = plan.nstream, nwave = plan.nwave, ncold = plan.ncold;
  for (int batch = 0;; ++batch) {
    // first batch: what the previous traversal of this graph needed (sources differ, level structure hardly)
    // (later batches: a traversal deeper than the graph's last ones -- twice the slots every time, up to 32: a host look costs
    //  what a dozen idle launches do)
    int nslots = batch == 0 ? bfs_class_slots(st, cls) : (st.levels_per_sync << (batch - 1 < 4 ? batch - 1 : 4));
    if (nslots > bfs_fused_state_t::EV_POOL / 3) nslots = bfs_fused_state_t::EV_POOL / 3;
    if (batch_events) MGX_HIP(hipEventRecord(st.ev0, s));
    const int first_slot = slot;
    for (int i = 0; i < nslots; ++i, ++slot) {
      const int arg = bfs_slot_arg(slot);
      if (a.chain_big_edges && slot > 0 && slot >= st.tail_from && opt.tail_chain && opt.tail_front && !plan.minis) chain_inplace(slot);
      const bool in_pool = 3 * i + 2 < bfs_fused_state_t::EV_POOL;
      const bool timed = st.time_kernels == 1 && in_pool;
      const bool timed_merged = st.time_kernels == 2 && in_pool && opt.merged && !lab_flags;
      if (timed_merged) {
        // the product launch itself between two events: its average duration is what rocprofv3 --stats reports for
        // k_bfs_push<., 0> too (launches of small or empty slots included on both sides)
        MGX_HIP(hipEventRecord(st.wev[3 * i], s));
        bfs_launch_push_part<0>(a, arg, ctx, coldt, nstream + ncold + nwave, nstream);
        MGX_HIP(hipEventRecord(st.wev[3 * i + 1], s));
      } else if (timed || !opt.merged || lab_flags) {
        // the parts as launches of their own: opener / chain, long rows, short rows
        bfs_launch_push_part<1>(a, arg, ctx, coldt, 1, 0);
        if (timed) MGX_HIP(hipEventRecord(st.wev[3 * i], s));
        if (a.long_min > 0) {
#ifdef MGX_LAB
          if (a.flags) hipLaunchKernelGGL(k_bfs_push_stream_diag, dim3(nstream), dim3(1024), bfs_push_lds_bytes(), s, a, arg);
          else
#endif
          bfs_launch_push_part<2>(a, arg, ctx, coldt, nstream + ncold, nstream);
        }
        if (timed) MGX_HIP(hipEventRecord(st.wev[3 * i + 1], s));
        bfs_launch_push_part<3>(a, arg, ctx, coldt, nwave, 0);
        if (timed) MGX_HIP(hipEventRecord(st.wev[3 * i + 2], s));
      } else {
        bfs_launch_push_part<0>(a, arg, ctx, coldt, nstream + ncold + nwave, nstream);
      }
      if (mode == 1 && !a.merged_pull)
        hipLaunchKernelGGL(k_bfs_pull_level<256>, dim3(ctx.num_cus * 8), dim3(256), 0, s, a, arg);
      bfs_enqueue_build(st, plan, arg, s);
      if (timed_merged) MGX_HIP(hipEventRecord(st.wev[3 * i + 2], s));    // (timing mode 2: the slot's queue build too)
    }
    if (plan.minis && batch == 0) { bfs_enqueue_mini(plan, slot, s); ++slot; }                // (the mid-size level behind the peak)
    if (a.chain_big_edges && (slot >= st.tail_from || plan.minis) && opt.tail_chain) chain_inplace(slot);    // (the stragglers: may end the traversal here)
    if (batch_events) MGX_HIP(hipEventRecord(st.ev1, s));
    // one read-back per batch: the counters and the first 64 trace slots (the flag alone would cost the same trip)
    constexpr size_t head_bytes = offsetof(bfs_ctrl_t, trace) + 64 * sizeof(u64);
    if ((opt.spin >= 0 ? opt.spin != 0 : st.spin) && !batch_events) {
      const u64 seq = ++st.seq;
      hipLaunchKernelGGL(k_bfs_publish, dim3(1), dim3(256), 0, s, (const bfs_ctrl_t*)st.ctrl.data(), st.host_ctrl, st.host_seq, seq,
                         (int)(head_bytes / 4));
      MGX_CHECK_LAUNCH("fused BFS: kernel launch");
      volatile u64* const flag = st.host_seq;
      long long spins = 0;
      while (*flag != seq) {
        if (++spins > 20000000LL) { MGX_HIP(hipStreamSynchronize(s)); break; }     // (a failed launch: let the runtime report it)
        __builtin_ia32_pause();
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    } else {
      MGX_CHECK_LAUNCH("fused BFS: kernel launch");
      MGX_HIP(hipMemcpyAsync(st.host_ctrl, st.ctrl.data(), head_bytes, hipMemcpyDeviceToHost, s));
      MGX_HIP(hipStreamSynchronize(s));
    }
    float ms = 0.f;
    if (batch_events) MGX_HIP(hipEventElapsedTime(&ms, st.ev0, st.ev1));
    st.level_kernel_ms += ms;
    for (int i = 0; st.time_kernels && i < nslots && 3 * i + 2 < bfs_fused_state_t::EV_POOL; ++i) {
      const int sl = first_slot + i;
      float wms = 0.f;
      if (st.time_kernels == 2) {
        if (!(opt.merged && !lab_flags)) break;
        MGX_HIP(hipEventElapsedTime(&wms, st.wev[3 * i], st.wev[3 * i + 1]));
        st.stream_kernel_ms += wms;
        st.stream_kernel_launches += 1;
        float bms = 0.f;                                  // ... and the queue build behind it (reported in the "wave" half)
        MGX_HIP(hipEventElapsedTime(&bms, st.wev[3 * i + 1], st.wev[3 * i + 2]));
        st.wave_kernel_ms += bms;
        st.wave_kernel_launches += 1;
        if (sl < 64) { st.level_stream_ms[sl] = wms; st.level_wave_ms[sl] = bms; }
        continue;
      }
      MGX_HIP(hipEventElapsedTime(&wms, st.wev[3 * i + 1], st.wev[3 * i + 2]));
      st.wave_kernel_ms += wms;
      st.wave_kernel_launches += 1;
      if (sl < 64) st.level_wave_ms[sl] = wms;
      if (a.long_min > 0) {
        MGX_HIP(hipEventElapsedTime(&wms, st.wev[3 * i], st.wev[3 * i + 1]));
        st.stream_kernel_ms += wms;
        st.stream_kernel_launches += 1;
        if (sl < 64) st.level_stream_ms[sl] = wms;
      }
    }
    if (st.batches < 256) st.batch_ms[st.batches++] = ms;
    st.level_kernel_launches += nslots;
    if (st.host_ctrl->done) break;
    // all `slot` slots launched so far have run; if the last one left both queues empty the traversal is over (no need
    // to launch the slot that would find that out)
    const u64 next = st.host_ctrl->cursor[slot % 3] | st.host_ctrl->lcursor[slot % 3];
    if ((next >> BFS_VSHIFT) == 0) {
      st.host_ctrl->done = 1;
      st.host_ctrl->levels = st.host_ctrl->slot_level[slot & 3];
      break;
    }
  }
  st.slots_used = slot;
  const int lv = st.host_ctrl->levels < BFS_MAX_TRACE ? st.host_ctrl->levels : BFS_MAX_TRACE;
  if (lv > 64) {                // the rest of the per-level trace (deep traversals only)
    MGX_HIP(hipMemcpyAsync(st.host_ctrl->trace + 64, st.ctrl.data()->trace + 64, (size_t)(lv - 64) * sizeof(u64), hipMemcpyDeviceToHost, s));
    MGX_HIP(hipStreamSynchronize(s));
  }
  bfs_learn_slots(st, a, st.host_ctrl, mode, BFS_MAX_TRACE, plan.minis, cls);
}

// COUNT traversals enqueued back to back with ONE host wait at the end (mgx_bfs_run_many): every traversal is complete --
// its labels written, its counters taken -- before the next one's init kernel resets the state, but the host does not
// look in between, so the ~18 us between the publish of one traversal and the first kernel of the next (the host's
// wake-up, the call, the first launch) and the publish kernel itself disappear.  Each traversal gets the slots the last
// traversals of the graph needed plus one; the head of its control block is copied to heads[i] by the NEXT traversal's
// init kernel (the last one's by k_bfs_publish).  A traversal that did not finish within its slots (a source with an
// unusual level structure) is run again on its own afterwards -- then the last source too, so that labels[] always holds
// the LAST source's traversal when the call returns.  heads: pinned host memory, bfs_many_head_bytes() apart.
constexpr size_t bfs_head_bytes() { return offsetof(bfs_ctrl_t, trace) + 64 * sizeof(u64); }
constexpr size_t bfs_many_head_bytes() { return (bfs_head_bytes() + 63) & ~(size_t)63; }
inline bfs_ctrl_t* bfs_many_head(char* heads, int i) { return (bfs_ctrl_t*)(heads + (size_t)i * bfs_many_head_bytes()); }

// Tried for deep graphs and dropped (round 5; a 2048 x 2048 grid, 3 690 levels of <= 8 192 edges): an M launch that GOES ON with the level
// it produced while that one is mid-size too -- 64 co-resident workgroups, a grid barrier (device-scope release / acquire) between
// two levels instead of a [push, build] pair of launches.  It ran the whole traversal in one launch (3 launch slots instead of
// 2 249) and took 60.2 ms against 61.5: ~16 us per level either way.  A level's cost there is its own chain of ~10 dependent
// memory round trips (cursor, queue entry, row, bitmap word, claim, row extent, cursor add, queue store), not the launches; on
// RMAT-22 the barrier made the stragglers' levels 1.5 % slower than the chain launch behind the M launch (0.3132 / 0.3086 ms).
// Also round 5, for the same graphs and for R-MAT's stragglers: a device-wide slot that finds a MID-SIZE level runs it M-launch style in
// the first 64 workgroups of its push launch (the build returns at once).  RMAT-22: 0.3098 against 0.3076 ms (the push kernel went from
// 18 to 48 spilled SGPRs); the grid: 68.7 against 62.7 ms -- 64 workgroups claiming by device-scope atomics are no faster per level
// than the [push, build] pair they replace.
// Tried for the batch and dropped (each measured on RMAT-22, 64 sources; the code is in the history of this file):
//   * two LANES -- state + HIP stream each -- with the sources alternating between them, so that the single-workgroup
//     launches at the start and end of one traversal overlap the device-wide launches of the other: 0.446 ms per traversal
//     against 0.349 (0.615 against 0.340 without M launches): two push launches that each want two 80 KB workgroups per CU
//     and the L2 for their bitmaps take turns instead of overlapping;
//   * two STATES on one stream with the host one traversal ahead, every traversal sized by the source that needs FEWEST slots
//     and an unfinished one continued behind the next instead of run again: 0.366 against 0.343 ms -- 35 of the 64 sources
//     needed the continuation, and each one drains the queue while the host looks at its control block;
//   * two lanes STAGGERED -- the device-wide slots of traversal i wait (hipStreamWaitEvent) for those of traversal i - 1, only
//     the one-workgroup launches at the two ends of a traversal (~58 of its 339 us) may run beside the other lane's big ones:
//     0.413 against 0.339 ms.  They do not run BESIDE a push launch -- it holds every CU's LDS and all 32 wave slots -- they
//     wait for it (k_bfs_chain_inplace 121 us, k_bfs_fused_init 62 us, k_bfs_mini 103 us in the trace) and then hold up their
//     own lane.  Reserving CUs for them with CU-masked streams works as such, but a grid sized for the masked device is dealt
//     round-robin to XCDs and shader engines whatever the mask left of each: 14 of 496 workgroups start a whole workgroup run
//     late (tools/cumask_probe2.hip), which doubles a launch of equal static shares; a symmetric reservation costs 32 CUs.
//     (profiles/r03/lanes_and_cu_masks.txt)
inline int bfs_fused_run_many(bfs_fused_state_t& st, const int* row_offsets, const int* col_indices, int* labels,
                              const int* srcs, int count, standard_context_t& ctx, char* heads, const bfs_layout_t* layout = nullptr,
                              int mode = 0, float alpha = 0.f, const int* in_offsets = nullptr, const int* in_indices = nullptr) {
  if (count <= 0) return 0;
  hipStream_t s = ctx.stream();
  // a traversal of the batch run on its own (a deep graph, a re-run): its row of the sources' shapes becomes row 0 of that call's
  auto run_one = [&](int i) {
    bfs_layout_t one;
    if (layout) { one = *layout; if (one.src_shapes) one.src_shapes += (size_t)i * 4; }
    bfs_fused_run(st, row_offsets, col_indices, labels, srcs[i], ctx, layout ? &one : nullptr, mode, alpha, in_offsets, in_indices);
  };
  const bfs_launch_plan_t plan = bfs_fused_plan(st, row_offsets, col_indices, labels, ctx, layout, mode, alpha, in_offsets, in_indices);
  const bfs_fused_args_t& a = plan.a;
  constexpr int head_words = (int)(bfs_head_bytes() / 4);
  // slots per traversal: what the last traversals of the graph needed at most.  A direction-optimising run gets one more, and
  // another one when its last traversals did not all need the same: where a source switches direction moves its level
  // structure (4 .. 7 slots by source at alpha = 1 on RMAT-22), a spare slot is two idle launches, ~5-9 us, and a traversal that
  // does not finish is run again.  Measured at alpha = 64, 16 sources: RMAT-24 9 re-runs and 1.28 ms per traversal -> 0 and
  // 0.78 (one call per source: 0.79), RMAT-23 0.47 -> 0.41, RMAT-20 0.19 -> 0.15; alpha = 1 on RMAT-22 0.385 -> 0.344.  Top-down
  // runs never needed it (0.3405 -> 0.3445 ms with it).
  bool uneven = false;
  for (int i = 1; i < 4 && i < st.recent_at; ++i) uneven |= st.recent_need[i] != st.recent_need[0];
  // ... and whatever mode: a batch that had to run a traversal again earns the handle's next batches a spare slot (at most 4),
  // taken back after eight batches in a row without a re-run (graphs whose sources differ in depth: R-MAT 16, 3 of 32 sources)
  int nslots = st.slots_hint + st.opts.many_spare + (mode == 1 ? (uneven ? 2 : 1) : 0) + st.auto_spare;
  if (nslots > 30) {
    // a DEEP graph (its last traversals needed more device-wide slots than a batch entry gets -- a grid, a road network): every
    // traversal would run out of slots and be run again on its own.  One call per source then, each with its own batches of
    // slots (round 5: grid2d-22, 79.9 -> 61.7 ms per traversal); the heads are what those calls leave.
    for (int i = 0; i < count; ++i) {
      run_one(i);
      memcpy(bfs_many_head(heads, i), st.host_ctrl, bfs_head_bytes());
    }
    return 0;
  }
  if (nslots < 1) nslots = 1;
  const int saved_tail = st.tail_from;
  st.tail_from = plan.minis ? (1 << 30) : nslots - 1;      // (no M launches: a chain launch in front of the last slot and behind the batch)
  const bool tail = a.chain_big_edges && st.opts.tail_chain;
  std::vector<int> last_slots((size_t)count, nslots), classes((size_t)count, BFS_SRC_UNKNOWN);   // per traversal: the slot the chain behind it works in (where an unfinished one stands), its class
  // the chain behind a traversal's last slot may run levels up to the list capacity here (a lone workgroup needs ~4.3 us per
  // 1000 edges: slower than a slot above ~4000 edges, but far cheaper than running the whole traversal again)
  bfs_launch_plan_t plan_tail = plan;
  plan_tail.a.chain_big_edges = BFS_CHAIN_CAP_BIG;
  for (int i = 0; i < count; ++i) {
    // (the head of the control block as the previous traversal left it goes to the host before the init kernel resets it)
    // (per source: which of the launches in front of the device-wide slots will find work, and how many slots its class needs)
    const int cls = bfs_classify_source(st, plan, layout, srcs[i], i);
    int my_slots = cls == BFS_SRC_UNKNOWN ? nslots : bfs_class_slots(st, cls) + st.opts.many_spare + st.auto_spare;
    if (my_slots > 30) my_slots = 30;
    if (my_slots < 1) my_slots = 1;
    int sl = bfs_enqueue_start(st, plan, srcs[i], ctx, i > 0 ? st.ctrl.data() : (const bfs_ctrl_t*)nullptr,
                               i > 0 ? bfs_many_head(heads, i - 1) : (bfs_ctrl_t*)nullptr, head_words, cls != BFS_SRC_SKIP);
    for (int k = 0; k < my_slots; ++k, ++sl) bfs_enqueue_slot(st, plan, sl, ctx);
    if (plan.minis) { bfs_enqueue_mini(plan, sl, s); ++sl; }
    if (tail) bfs_enqueue_chain_inplace(plan_tail, sl, s);
    last_slots[(size_t)i] = sl;
    classes[(size_t)i] = cls;
  }
  st.tail_from = saved_tail;
  const u64 seq = ++st.seq;
  hipLaunchKernelGGL(k_bfs_publish, dim3(1), dim3(256), 0, s, (const bfs_ctrl_t*)st.ctrl.data(), bfs_many_head(heads, count - 1), st.host_seq, seq, head_words);
  MGX_CHECK_LAUNCH("fused BFS (batch of sources): kernel launch");
  {
    volatile u64* const flag = st.host_seq;
    long long spins = 0;
    while (*flag != seq) {
      if (++spins > 20000000LL) { MGX_HIP(hipStreamSynchronize(s)); break; }
      __builtin_ia32_pause();
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
  }
  // finished?  (as in bfs_fused_run: done, or the last slot left both queues of the next one empty)
  int reruns = 0;
  bool redo_last = false;
  for (int i = 0; i < count; ++i) {
    bfs_ctrl_t* const h = bfs_many_head(heads, i);
    const int last_slot = last_slots[(size_t)i];
    if (!h->done) {
      const u64 next = h->cursor[last_slot % 3] | h->lcursor[last_slot % 3];
      if ((next >> BFS_VSHIFT) == 0) { h->done = 1; h->levels = h->slot_level[last_slot & 3]; }
    }
    if (h->done) { bfs_learn_slots(st, a, h, mode, 64, plan.minis, classes[(size_t)i]); continue; }
    run_one(i);
    memcpy(h, st.host_ctrl, bfs_head_bytes());
    ++reruns;
    if (i != count - 1) redo_last = true;
  }
  if (redo_last) run_one(count - 1);
  st.slots_used = last_slots[(size_t)count - 1];
  if (reruns > 0) { if (st.auto_spare < 4) ++st.auto_spare; st.clean_batches = 0; }
  else if (st.auto_spare > 0 && ++st.clean_batches >= 8) { --st.auto_spare; st.clean_batches = 0; }
  return reruns;
}

}  // namespace mgx
