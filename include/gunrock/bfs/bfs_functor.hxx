// gunrock/bfs/bfs_functor.hxx -- the device functor the operators inline for BFS.
// Same static members and semantics as the reference's bfs_functor_t
// (gunrock/src/bfs/bfs_functor.hxx:7-53):
//   cond_filter            keep anything that is not the -1 "loser" marker        (:9-11)
//   cond_uniq              idempotent-mode relabel                                 (:13-24)
//   cond_advance           destination still unlabelled                            (:26-28)
//   apply_advance          claim it: CAS labels[dst] -1 -> iteration+1             (:30-33)
//   cond_sparse_to_dense   vertex belongs to the level being turned into a bitmap  (:35-37)
//   cond_gen_unvisited     vertex still unlabelled                                 (:39-41)
//   get_value_to_reduce / write_reduced_value   unused by BFS, kept for the concept (:43-51)
#pragma once
#include "bfs_problem.hxx"

namespace gunrock {
namespace bfs {

struct bfs_functor_t {
  using slice_t = bfs_problem_t::data_slice_t;

  // ---- advance: claim unlabelled destinations for level iteration + 1 -------------------------------------------
  static __device__ __forceinline__ bool cond_advance(int, int dst, int, int, int, slice_t* d, int) {
    return d->d_labels[dst] == -1;
  }
  static __device__ __forceinline__ bool apply_advance(int, int dst, int, int, int, slice_t* d, int iteration) {
    // The operators call this for EVERY edge (advance.hxx:57-58).  A label only ever goes from -1 to a level, so a
    // plain read that does not see -1 already is the answer the CAS would give; device-scope atomics run at the
    // memory side on MI355X (~25 G/s), and nine edges in ten of an R-MAT traversal point at labelled vertices.
    // (Round 5, tried and dropped: a visited BITMAP beside the labels -- 512 KB, L2-resident -- probed before the label: 2.49 against
    //  2.21 ms per RMAT-22 traversal on the operator path.  Within a level the bit of a vertex another XCD has just labelled is not
    //  in this XCD's L2 yet, so the level that carries the edges pays both probes for two edges in five.)
    int* const label = d->d_labels + dst;
    return *label == -1 && atomicCAS(label, -1, iteration + 1) == -1;
  }

  // ---- filter: drop the slots of the edges that lost -------------------------------------------------------------
  static __device__ __forceinline__ bool cond_filter(int slot_value, slice_t*, int) { return slot_value != -1; }
  // (the test looks at the slot's value alone: the advance that writes the slot may evaluate it, advance.hxx / filter.hxx)
  static constexpr bool cond_filter_of_slot_value_only = true;

  // idempotent mode (the reference's uniquify path): relabel unless the vertex already carries a level of this or
  // an earlier superstep; vertex 0 is skipped there too (bfs_functor.hxx:13-24)
  static __device__ __forceinline__ bool cond_uniq(int v, slice_t* d, int iteration) {
    if (v <= 0) return false;
    const int level = d->d_labels[v];
    if (level > 0 && level <= iteration) return false;
    d->d_labels[v] = iteration + 1;
    return true;
  }

  // ---- pull phase ------------------------------------------------------------------------------------------------
  static __device__ __forceinline__ bool cond_sparse_to_dense(int v, slice_t* d, int iteration) {
    return d->d_labels[v] == iteration;
  }
  static __device__ __forceinline__ bool cond_gen_unvisited(int v, slice_t* d, int) { return d->d_labels[v] == -1; }

  // ---- unused by BFS, part of the functor concept ------------------------------------------------------------------
  static __device__ __forceinline__ int get_value_to_reduce(int, slice_t*, int iteration) { return iteration; }
  static __device__ __forceinline__ void write_reduced_value(int, int, slice_t*, int) {}
};

// The functor of the IDEMPOTENT traversal (advance<idempotence = true> + uniquify_kernel, advance.hxx:60,
// filter.hxx:95-119): advance claims nothing -- every neighbour goes to the output, no atomics on labels -- and the label
// is written by cond_uniq, for the one edge per vertex that uniquify's exact bitmask cull lets through.  cond_uniq here is
// what upstream's (bfs_functor.hxx:13-24, restated above with its quirks) means to be: it does not skip vertex 0 and
// does not relabel the source when an edge leads back to it.
struct bfs_idempotent_functor_t : bfs_functor_t {
  static __device__ __forceinline__ bool cond_advance(int, int, int, int, int, slice_t*, int) { return true; }
  static __device__ __forceinline__ bool apply_advance(int, int, int, int, int, slice_t*, int) { return true; }
  static __device__ __forceinline__ bool cond_uniq(int v, slice_t* d, int iteration) {
    if (v < 0 || d->d_labels[v] != -1) return false;
    d->d_labels[v] = iteration + 1;
    return true;
  }
};

}  // namespace bfs
}  // namespace gunrock
