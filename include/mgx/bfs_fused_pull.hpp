// mgx/bfs_fused_pull.hpp -- bottom-up level of direction-optimising runs.
// One lane per vertex: an unvisited vertex walks its in-edges until it meets a member of the level's
// frontier bitmap (early exit -- the reference's advance_backward_kernel inspects every in-edge,
// advance.hxx:142-157).  Like the push kernels it only sets mark[v]; k_bfs_build turns the marks into bitmap
// bits, labels and the next level's sizes.
#pragma once
#include "bfs_fused.hpp"

namespace mgx {

template <int NT>
__global__ __launch_bounds__(NT) void k_bfs_pull_level(bfs_fused_args_t a, int arg) {
  __shared__ unsigned long long s_insp;
  bfs_ctrl_t* const c = a.ctrl;
  if (c->done || !c->pull) return;                 // the level's opener (push launch): termination and direction
  const int n = a.n;
  long long per_v = ((long long)n + gridDim.x - 1) / gridDim.x;
  per_v = (per_v + NT - 1) / NT * NT;
  const long long v_begin = (long long)blockIdx.x * per_v;
  if (v_begin >= n) return;
  const long long v_end = (v_begin + per_v < n) ? v_begin + per_v : n;
  if (threadIdx.x == 0) s_insp = 0ull;
  __syncthreads();
  const int lane = lane_id();
  int inspected = 0;
  for (long long base = v_begin; base < v_end; base += NT) {
    const long long v = base + threadIdx.x;
    const bool active = v < v_end;
    if (active) {
      const u32 word = a.visited[v >> 5];
      if (!((word >> (v & 31)) & 1u)) {
        const u32 e0 = a.in_offsets[v], e1 = a.in_offsets[v + 1];
        for (u32 e = e0; e < e1; ++e) {
          const u32 u = (u32)a.in_indices[e];
          ++inspected;
          if ((a.frontier_bits[u >> 5] >> (u & 31)) & 1u) { a.mark[v] = 1; break; }
        }
      }
    }
  }
  const int insp = wave_sum(inspected);
  if (lane == 0 && insp) atomicAdd(&s_insp, (unsigned long long)insp);
  __syncthreads();
  if (threadIdx.x == 0 && s_insp) atomicAdd(&c->pull_edges, (u64)s_insp);
}

}  // namespace mgx
