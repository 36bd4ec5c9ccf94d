// mgx/bfs_fused_run.hpp -- host driver of the fused BFS.  One init kernel, then per level
//   k_bfs_push_level (long rows streamed + short rows searched, one grid)  [-> k_bfs_pull_level]  -> k_bfs_build
// launched back to back for as many levels as the previous traversal of the graph had; the host reads the control
// block back once per batch.  Two launch schemes (see bfs_fused_run): "direct" with the level number as a kernel
// argument, and "slots" with k_bfs_small_levels in front of every level and the level counter on the device.
#pragma once
#include "bfs_fused.hpp"
#include "bfs_fused_pull.hpp"
#include "bfs_fused_small.hpp"
#include "bfs_fused_stream.hpp"
#include "bfs_fused_wave.hpp"

namespace mgx {

// layout (optional): a hub-first relabelled copy of the CSR plus the two id maps; labels stay in the
// original id space either way.
struct bfs_layout_t {
  const int* row_offsets = nullptr;
  const int* col_indices = nullptr;
  const int* new_of_old = nullptr;
  const int* old_of_new = nullptr;
};

// Template instances of the two push kernels.  cold_test: probe the bitmap word of neighbours outside the LDS
// prefix (big graphs: many cold endpoints) or mark them untested (k_bfs_build tests the bitmap anyway).
constexpr int BFS_STREAM_HOTW2 = 20400;   // two workgroups per CU: 80 KB of bitmap each

// Both push kernels of a level in ONE launch: the first `nstream` workgroups run the streaming body over the
// long-row queue, the others the wave body over the short-row queue (default shapes of the two kernels above).
// Saves a launch per level (~6 us of device time each, measured) and lets the short rows start while the last
// slices of the long rows drain.  Profiling runs (bfs_fused_state_t::time_kernels) launch the two kernels
// separately so that each can be bracketed by events.
// open_here (direct scheme, see bfs_fused_run: explicit level numbers, no k_bfs_small_levels in front; 2: a rank of a
// partitioned run): the level's bookkeeping is done by one thread of this grid.  Nothing it writes is read by the level's own kernels in a
// top-down run: they take the queue sizes from the cursors, which the previous level's k_bfs_build completed.
template <bool COLDT>
__global__ __launch_bounds__(1024, 8) void k_bfs_push_level(bfs_fused_args_t a, int level, u32 nstream, int open_here) {
  if (open_here && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) bfs_begin_level(a, level, open_here == 2);
  if (blockIdx.x < nstream) bfs_stream_body<1024, BFS_STREAM_HOTW2, 8, COLDT, false, true>(a, level, blockIdx.x, nstream);
  else bfs_wave_body<1024, 18000, COLDT, false>(a, level, blockIdx.x - nstream, gridDim.x - nstream);
}

inline void bfs_set_kernel_attributes() {
  static bool attr_set = false;
  if (attr_set) return;
#define MGX_SET_LDS(K_) MGX_HIP(hipFuncSetAttribute((const void*)K_, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
  MGX_SET_LDS((k_bfs_small_levels<BFS_SMALL_NT>));
  MGX_SET_LDS(k_bfs_push_level<false>);
  MGX_SET_LDS(k_bfs_push_level<true>);
  MGX_SET_LDS((k_bfs_push_level_wave<1024, 18000, false>));
  MGX_SET_LDS((k_bfs_push_level_wave<1024, 18000, true>));
  MGX_SET_LDS((k_bfs_push_level_stream<1024, BFS_STREAM_HOTW2, 8, false, true, true>));
  MGX_SET_LDS((k_bfs_push_level_stream<1024, BFS_STREAM_HOTW2, 8, false, false, true>));
  MGX_SET_LDS((k_bfs_push_level_stream<1024, BFS_STREAM_HOTW2, 8, true, false, true>));
#undef MGX_SET_LDS
  attr_set = true;
}

// -1: decide by size (cold test when the bitmap is at least 8 x the LDS prefix), 0 / 1: forced (MGX_BFS_COLD_TEST)
inline bool bfs_cold_test(int n) {
  const char* const e = getenv("MGX_BFS_COLD_TEST");               // (read per call: the tests switch it)
  const int forced = e ? atoi(e) : -1;
  if (forced >= 0) return forced != 0;
  return (long long)n >= 8ll * 32 * BFS_STREAM_HOTW;
}

// The two push kernels as launches of their own (profiling runs, MGX_BFS_MERGED_PUSH=0, the instrumented build).
// Shapes that were measured and dropped, RMAT-22 (stream kernel of the big level / whole BFS at the time):
//   stream  2 workgroups x 1024 threads per CU = 32 waves, 80 KB of bitmap each, 8 non-temporal loads per lane (kept):
//           167 us / 0.66 ms; 16 loads per lane 190 / 0.68; 1 workgroup per CU with 160 KB of bitmap 213-217 / 0.69-0.71;
//           cached instead of non-temporal col_indices loads 0.582 vs 0.565 ms per traversal; 16-byte-per-lane reads
//           (sub-rounds of 256 consecutive edges) 190-245 us; on partitioned RMAT-25 (cold test) one workgroup per CU
//           with 160 KB was 3 % faster than two with 80 KB -- not enough for a second merged shape;
//   wave    2 x 1024 threads per CU (kept) 67 us, 2 x 512 threads with more bitmap in LDS 82 us; non-temporal loads: same;
//   build   512 threads (kept) 0.445 ms per traversal, 256: 0.453-0.461, 1024: 0.476.
inline void bfs_launch_stream(const bfs_fused_args_t& a, int level, standard_context_t& ctx) {
  hipStream_t s = ctx.stream();
  if (a.long_min <= 0) return;
  const size_t lds2 = bfs_stream_lds_bytes(BFS_STREAM_HOTW2);
  if (bfs_cold_test(a.n))
    hipLaunchKernelGGL((k_bfs_push_level_stream<1024, BFS_STREAM_HOTW2, 8, true, false, true>), dim3(ctx.num_cus * 2), dim3(1024), lds2, s, a, level);
  else if (a.flags)     // MGX_BFS_FLAGS set: the instrumented build
    hipLaunchKernelGGL((k_bfs_push_level_stream<1024, BFS_STREAM_HOTW2, 8, false, true, true>), dim3(ctx.num_cus * 2), dim3(1024), lds2, s, a, level);
  else                  // col_indices are read once: non-temporal loads leave L2 to the bitmap, marks and queues
    hipLaunchKernelGGL((k_bfs_push_level_stream<1024, BFS_STREAM_HOTW2, 8, false, false, true>), dim3(ctx.num_cus * 2), dim3(1024), lds2, s, a, level);
}

inline void bfs_launch_wave(const bfs_fused_args_t& a, int level, standard_context_t& ctx) {
  hipStream_t s = ctx.stream();
  const size_t lds = bfs_wave_lds_bytes(1024, 18000);
  if (bfs_cold_test(a.n)) hipLaunchKernelGGL((k_bfs_push_level_wave<1024, 18000, true>), dim3(ctx.num_cus * 2), dim3(1024), lds, s, a, level);
  else hipLaunchKernelGGL((k_bfs_push_level_wave<1024, 18000, false>), dim3(ctx.num_cus * 2), dim3(1024), lds, s, a, level);
}

inline void bfs_launch_push(const bfs_fused_args_t& a, int level, standard_context_t& ctx, int open_here = 0) {
  const char* const merged_str = getenv("MGX_BFS_MERGED_PUSH");      // (read per call: the tests switch it)
  const int merged = merged_str ? atoi(merged_str) : 1;
  if (!merged || a.flags) {
    if (open_here) hipLaunchKernelGGL(k_bfs_level_begin, dim3(1), dim3(64), 0, ctx.stream(), a, level, open_here == 2 ? 1 : 0);
    bfs_launch_stream(a, level, ctx);
    bfs_launch_wave(a, level, ctx);
    return;
  }
  hipStream_t s = ctx.stream();
  const size_t lds_s = bfs_stream_lds_bytes(BFS_STREAM_HOTW2), lds_w = bfs_wave_lds_bytes(1024, 18000);
  const size_t lds = lds_s > lds_w ? lds_s : lds_w;
  const u32 nstream = a.long_min > 0 ? (u32)ctx.num_cus * 2 : 0u;
  const u32 nwave = (u32)ctx.num_cus * 2;
  if (bfs_cold_test(a.n))
    hipLaunchKernelGGL(k_bfs_push_level<true>, dim3(nstream + nwave), dim3(1024), lds, s, a, level, nstream, open_here);
  else
    hipLaunchKernelGGL(k_bfs_push_level<false>, dim3(nstream + nwave), dim3(1024), lds, s, a, level, nstream, open_here);
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
  bfs_fused_args_t a;
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
  a.flags = 0;
  a.count_marks = (st.count_marks || st.time_kernels) ? 1 : 0;
  if (const char* e = getenv("MGX_BFS_FLAGS")) a.flags = atoi(e);
  const long long nwords = ((long long)st.n + 31) / 32;
  hipLaunchKernelGGL(k_bfs_fused_init, dim3(grid_for(((long long)st.n + 3) / 4, BLOCK, ctx.num_cus * 8)), dim3(BLOCK), 0, s, a, src, nwords);
  st.level_kernel_ms = 0.0;
  st.level_kernel_launches = 0;
  st.wave_kernel_ms = 0.0;
  st.wave_kernel_launches = 0;
  st.stream_kernel_ms = 0.0;
  st.stream_kernel_launches = 0;
  st.batches = 0;
  // Two launch schemes.
  //   slots  [k_bfs_small_levels, push, (pull), build] with the level counter on the device: the single-workgroup
  //          kernel runs any number of small levels inside one launch (deep graphs: hundreds of tiny levels) and
  //          opens the next big one; on a shallow graph it is an idle ~5 us launch in front of every big level.
  //   direct [push, (pull), build] per level with the level number as an argument; the level's bookkeeping rides on
  //          the push launch (the direction of a direction-optimising level too: bfs_level_pulls).
  // The scheme follows the previous traversal of the graph: direct unless that one was deep (MGX_BFS_DIRECT forces).
  const char* const direct_str = getenv("MGX_BFS_DIRECT");          // (read per run: the tests switch it)
  const int direct_env = direct_str ? atoi(direct_str) : -1;
  // (profiling runs with events around the two push kernels keep the slot scheme unless forced: there the tiny levels
  //  stay inside the single-workgroup kernel instead of adding no-op launches to the kernels' averages)
  const bool direct = direct_env >= 0 ? direct_env != 0 : (st.direct_levels && !st.time_kernels);
  const bool batch_events = st.time_kernels || st.time_batches;    // (an event costs ~6 us of stream gap)
  int slot = 0;
  for (int batch = 0;; ++batch) {
    // first batch: what the previous traversal of this graph needed (sources differ, level structure hardly)
    int nslots = batch == 0 ? (direct ? st.levels_hint : st.slots_hint) : st.levels_per_sync;
    if (nslots > bfs_fused_state_t::EV_POOL / 3) nslots = bfs_fused_state_t::EV_POOL / 3;
    if (batch_events) MGX_HIP(hipEventRecord(st.ev0, s));
    for (int i = 0; i < nslots; ++i, ++slot) {
      const int lv_arg = direct ? slot : -1;
      if (!direct)
        hipLaunchKernelGGL(k_bfs_small_levels<BFS_SMALL_NT>, dim3(1), dim3(BFS_SMALL_NT), bfs_small_lds_bytes(), s, a,
                           st.small_max_edges);
      const bool timed = st.time_kernels && 3 * i + 2 < bfs_fused_state_t::EV_POOL;
      if (timed) {
        if (direct) hipLaunchKernelGGL(k_bfs_level_begin, dim3(1), dim3(64), 0, s, a, lv_arg, 0);
        MGX_HIP(hipEventRecord(st.wev[3 * i], s));
        bfs_launch_stream(a, lv_arg, ctx);
        MGX_HIP(hipEventRecord(st.wev[3 * i + 1], s));
        bfs_launch_wave(a, lv_arg, ctx);
        MGX_HIP(hipEventRecord(st.wev[3 * i + 2], s));
      } else {
        bfs_launch_push(a, lv_arg, ctx, direct ? 1 : 0);
      }
      if (mode == 1)
        hipLaunchKernelGGL(k_bfs_pull_level<256>, dim3(ctx.num_cus * 8), dim3(256), 0, s, a, lv_arg);
      hipLaunchKernelGGL((k_bfs_build<512, true>), dim3(bfs_build_grid(st.n, 512)), dim3(512), 0, s, a, lv_arg,
                         (const u32*)nullptr, labels, st.n, 1, 0, 1);       // (2 workgroups per CU overlap their phases)
    }
    if (batch_events) MGX_HIP(hipEventRecord(st.ev1, s));
    // one read-back per batch: the counters and the first 64 trace slots (the flag alone would cost the same trip)
    MGX_HIP(hipMemcpyAsync(st.host_ctrl, st.ctrl.data(), offsetof(bfs_ctrl_t, trace) + 64 * sizeof(u64), hipMemcpyDeviceToHost, s));
    MGX_HIP(hipStreamSynchronize(s));
    float ms = 0.f;
    if (batch_events) MGX_HIP(hipEventElapsedTime(&ms, st.ev0, st.ev1));
    st.level_kernel_ms += ms;
    for (int i = 0; st.time_kernels && i < nslots && 3 * i + 2 < bfs_fused_state_t::EV_POOL; ++i) {
      const int sl = slot - nslots + i;
      float wms = 0.f;
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
    if (direct) {
      // all `slot` levels launched so far have run; if the last build left both queues empty the traversal is over
      // (no need to launch the level that would find that out)
      const u64 next = st.host_ctrl->cursor[slot % 3] | st.host_ctrl->lcursor[slot % 3];
      if ((next >> BFS_VSHIFT) == 0) {
        st.host_ctrl->done = 1;
        st.host_ctrl->levels = slot;
        break;
      }
    }
  }
  if (direct) st.levels_hint = st.host_ctrl->levels > 0 ? st.host_ctrl->levels : 1;
  else st.slots_hint = st.host_ctrl->slots > 0 ? st.host_ctrl->slots : 1;
  st.direct_levels = st.host_ctrl->levels <= st.direct_max_levels;
  const int lv = st.host_ctrl->levels < BFS_MAX_TRACE ? st.host_ctrl->levels : BFS_MAX_TRACE;
  if (lv > 64) {                // the rest of the per-level trace (deep traversals only)
    MGX_HIP(hipMemcpyAsync(st.host_ctrl->trace + 64, st.ctrl.data()->trace + 64, (size_t)(lv - 64) * sizeof(u64), hipMemcpyDeviceToHost, s));
    MGX_HIP(hipStreamSynchronize(s));
  }
}

}  // namespace mgx
