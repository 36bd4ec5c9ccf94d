// mgx/bfs_fused_pull.hpp -- bottom-up level of direction-optimising runs.
// One lane per vertex: an unvisited vertex walks its in-edges until it meets a member of the level's
// frontier bitmap (early exit -- the reference's advance_backward_kernel inspects every in-edge,
// advance.hxx:142-157); what is left of a long list after a few fruitless probes is walked by the whole wave, 64 edges a step.  Like the push kernels it only sets mark[v]; k_bfs_build turns the marks into bitmap
// bits, labels and the next level's sizes.
#pragma once
#include "bfs_fused.hpp"

namespace mgx {

// the bottom-up sweep of workgroup `block` of `nblocks`; s_insp_p: one 64-bit word of LDS; all threads
template <int NT>
__device__ __forceinline__ void bfs_pull_body(const bfs_fused_args_t& a, u32 block, u32 nblocks, unsigned long long* s_insp_p) {
  unsigned long long& s_insp = *s_insp_p;
  bfs_ctrl_t* const c = a.ctrl;
  const int n = a.n;
  long long per_v = ((long long)n + nblocks - 1) / nblocks;
  per_v = (per_v + NT - 1) / NT * NT;
  const long long v_begin = (long long)block * per_v;
  if (v_begin >= n) return;
  const long long v_end = (v_begin + per_v < n) ? v_begin + per_v : n;
  if (threadIdx.x == 0) s_insp = 0ull;
  __syncthreads();
  const int lane = lane_id();
  int inspected = 0;
  for (long long base = v_begin; base < v_end; base += NT) {
    const long long v = base + threadIdx.x;
    const bool active = v < v_end;
    bool unvisited = false;
    u32 e0 = 0, e1 = 0;
    if (active) {
      const u32 word = a.visited[v >> 5];
      unvisited = !((word >> (v & 31)) & 1u);
      if (unvisited) { e0 = a.in_offsets[v]; e1 = a.in_offsets[v + 1]; }
    }
    // Every lane starts on its own list, serially, and is out at the first frontier member: in-edge lists are sorted by
    // neighbour id and the hub-first layout puts the vertices most likely to be in the frontier first, so most lanes
    // are done within a few probes.  A short list is walked to its end that way.  A LONG list whose first PULL_PROBES
    // entries held no frontier member (a lane walking a 10 000-entry list alone would hold its wave for as long) is
    // handed to the whole wave afterwards: 64 consecutive in-edges per step, coalesced, out at the first step in which
    // any lane meets a frontier member.  (Taking every long list cooperatively from its first entry was measured 3.5 x
    // slower on a level with many unvisited mid-degree vertices: 64 lists one after the other per wave, a dependent
    // load each, where 64 lanes would have been done after a probe or two each.)
    constexpr u32 PULL_PROBES = 8;
    bool found = false;
    u32 e = e0;
    if (unvisited) {
      const bool is_long = e1 - e0 >= (u32)WAVE;
      const u32 stop = is_long ? e0 + PULL_PROBES : e1;
      for (; e < stop; ++e) {
        const u32 u = (u32)a.in_indices[e];
        ++inspected;
        if ((a.frontier_bits[u >> 5] >> (u & 31)) & 1u) { found = true; break; }
      }
      if (found) a.mark[v] = 1;
    }
    u64 big = __ballot(unvisited && !found && e < e1);
    while (big) {
      const int leader = __ffsll((long long)big) - 1;
      big &= big - 1ull;
      const u32 s0 = (u32)__shfl((int)e, leader, WAVE), s1 = (u32)__shfl((int)e1, leader, WAVE);
      bool hit_any = false;
      for (u32 b = s0; b < s1 && !hit_any; b += WAVE) {
        const u32 ee = b + (u32)lane;
        bool hit = false;
        if (ee < s1) {
          const u32 u = (u32)a.in_indices[ee];
          ++inspected;
          hit = (a.frontier_bits[u >> 5] >> (u & 31)) & 1u;
        }
        hit_any = __ballot(hit) != 0ull;
      }
      if (hit_any && lane == leader) a.mark[v] = 1;
    }
  }
  const int insp = wave_sum(inspected);
  if (lane == 0 && insp) atomicAdd(&s_insp, (unsigned long long)insp);
  __syncthreads();
  if (threadIdx.x == 0 && s_insp) atomicAdd(&c->pull_edges, (u64)s_insp);
}

// ... as a launch of its own behind the push launch (MGX_BFS_MERGED_PULL=0; the product runs the body inside k_bfs_push:
// a launch that finds nothing to do costs 4.5-5.7 us, and a direction-optimising traversal had one per level)
template <int NT>
__global__ __launch_bounds__(NT) void k_bfs_pull_level(bfs_fused_args_t a, int arg) {
  __shared__ unsigned long long s_insp;
  (void)arg;
  const bfs_ctrl_t* const c = a.ctrl;
  if (c->done || !c->pull) return;                 // the level's opener (push launch): termination and direction
  bfs_pull_body<NT>(a, blockIdx.x, gridDim.x, &s_insp);
}

}  // namespace mgx
