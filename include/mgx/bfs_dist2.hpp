// mgx/bfs_dist2.hpp -- partitioned BFS, generation 2: every rank runs the FUSED level kernels
// (bfs_fused_stream.hpp / bfs_fused_wave.hpp) on its rows; ranks exchange dense "newly visited" bitmaps.
//
// Generation 1 (bfs_dist.hpp) expands with the operator-path scan + LBS kernels and ships per-owner id
// lists: ~10 GTEPS per rank on RMAT-22, a ninth of the single-GPU fused path.  Here:
//   * ids are GLOBAL and hub-first (renumbered by descending global degree, as in the single-GPU layout);
//     vertex v is owned by rank v % G (cyclic, so every rank holds its share of the hubs), local row v / G;
//   * every rank keeps the visited bitmap of ALL n vertices (n/8 bytes: 4 MB at RMAT-25) and a mark byte per
//     vertex.  A level runs the same push kernels as the single-GPU path on the rank's rows: bitmap test in
//     LDS/L2, plain byte stores into mark[] -- no atomics, no owner consulted;
//   * new_bits = marks & ~bitmap is the rank's discoveries of the level (any owner).  One all-gather of
//     these bitmaps per level replaces the id exchange: volume n/8 bytes per rank and level whatever the
//     frontier size, all 7 xGMI links busy, no counts to exchange first;
//   * merge: OR of the G bitmaps = the level's global discoveries; every rank ORs them into its bitmap (so
//     all ranks agree before the next level) and k_bfs_build turns the bits it OWNS into labels and into its
//     next local queues (row start, scanned degree; packed-cursor append as everywhere else).
//   * SPARSE levels do not ship bitmaps ("switch representation per level by density").  The sweep that turns
//     marks into new bits also compacts the new vertices into a short ID LIST (a header word with the count + up to
//     list_cap ids, list_cap = n / (256 ranks)); the lists are all-gathered first -- n / 64 bytes per rank instead of n / 8 --
//     and if every rank's discoveries fitted (the headers tell every rank the same), k_d2_lists_apply applies them:
//     atomicOr on the bitmap decides a vertex once, its owner labels it and appends it to its next queue (block scan,
//     one packed cursor atomic per workgroup).  Only when some rank overflowed does the level go through the bitmap
//     exchange below.  The host learns the decision (and whether the level found anything at all: the sum of the counts)
//     from three words in pinned memory it spins on -- the one host round trip of a level.
// Labels are the global BFS depths, identical to the single-GPU result.
//
// Round 4 ("the rank engine"; every item against a switch, MGX_DIST_* / MGX_BFS_COLD_PACK):
//   * a level of no more edges than the id list holds ids appends its discoveries to the list from the push launch itself
//     (bfs_fused_sparse.hpp) -- k_d2_newbits has nothing to sweep; a level whose push stored many marks declares its list
//     overflowed without filling it;
//   * the OR-merge of the ranks' maps runs inside the queue build (k_bfs_build2<., 2, RANKS>, 2 / 4 / 8 / 16 ranks), which also
//     leaves the frontier over the rank's LOCAL rows for the vertex-by-vertex walk of its short rows (bfs_fused_vshort.hpp, with
//     the cold test a bitmap of 2^26 vertices needs);
//   * the cold-edge pass reads four bytes per pair, its bitmaps and those of the deferred hot marks are ORed together by a stream
//     kernel in front of the sweep (k_d2_cold_reduce), and the unit blocks hold the rows' hot entries only, three bytes each;
//   * the rank's new-bit map is all zero between levels: the sweep writes every word, a sparse level ORs single bits, whoever
//     consumes the map (list merge: the words of the rank's own list; bitmap merge: all of it) clears what it read.
#pragma once
#include <cstddef>
#include <memory>

#include "bfs_fused_run.hpp"
#include "comm.hpp"
#include "env.hpp"

namespace mgx {

// out[w] = the rank's discoveries of the level: vertices it marked that are not in the bitmap.  list (optional): the same
// vertices as ids -- list[0] counts them (the caller zeroed it; exact up to list_cap, some value above it once the list has
// overflowed), list[D2_LIST_HEAD + i] holds the first list_cap.
// what k_d2_newbits needs of a rank's cold-edge pass (bfs_fused_cold.hpp): cold workgroup k of slice q left the vertices it
// discovered in [lo[q], lo[q] + BFS_COLD_WORDS * 32) as a bitmap in flush + k * BFS_COLD_WORDS
struct d2_cold_view_t {
  const u32* flush = nullptr;
  int slices = 0;
  int reduced = 0;                     // the first buffer of a slice holds the OR of all of them (k_d2_cold_reduce ran in front)
  u32 slice_n = 0;                     // vertices of a slice; slice number (v - lo[0]) / slice_n -> index into lo / wgs, 255: holds no pairs
  unsigned char qof[128] = {};
  u32 lo[BFS_COLD_MAX_SLICES] = {};
  u32 wgs[BFS_COLD_MAX_SLICES + 1] = {};
};
// The bitmaps the cold workgroups of a slice left behind (bfs_fused_cold.hpp), ORed into the slice's first one: 16 bytes per
// lane, eight buffers in flight -- a stream (a thousand buffers of 80 KB on RMAT-26 / 8: 83 MB).  Returns at once on a level
// whose push did not run the cold pass.  (Inside the push launch -- the last workgroup of a slice to finish does it -- the
// fences that make the other workgroups' stores visible across the XCDs' L2s tripled the launch: 368 -> 1 230 us.)
// The same for the bitmaps of the level's deferred hot marks (bfs_hot_epilogue: ctrl->flush_count[level & 1] buffers of
// BFS_FLUSH_WORDS words in defer_buf): the threads behind the slices'.
__global__ __launch_bounds__(BLOCK) void k_d2_cold_reduce(d2_cold_view_t cv, const bfs_ctrl_t* c, int level, u32* flush, u32* defer_buf) {
  if (bfs_d2_frozen(c, level)) return;
  if (c->d2_append_level == level) return;
  constexpr u32 Q = BFS_COLD_WORDS / 4;
  const u32 t = blockIdx.x * BLOCK + threadIdx.x;
  const u32 cold_threads = (u32)cv.slices * Q;
  u32 parts, i, stride4;
  uint4* first;
  if (t < cold_threads) {
    if (!cv.flush || c->cold_slot != level) return;
    const u32 q = t / Q;
    i = t % Q;
    parts = cv.wgs[q + 1] - cv.wgs[q];
    first = (uint4*)(flush + (size_t)cv.wgs[q] * BFS_COLD_WORDS);
    stride4 = Q;
  } else {
    i = t - cold_threads;
    if (!defer_buf || i >= (u32)BFS_FLUSH_WORDS / 4u) return;
    parts = c->flush_count[level & 1];
    if (parts > (u32)BFS_FLUSH_MAX) parts = (u32)BFS_FLUSH_MAX;
    first = (uint4*)defer_buf;
    stride4 = (u32)BFS_FLUSH_WORDS / 4u;
  }
  if (parts < 2u) return;
  const u32 Qs = stride4;
  uint4 acc = first[i];
  u32 k = 1;
  for (; k + 16u <= parts; k += 16u) {
    uint4 v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = first[(size_t)(k + (u32)j) * Qs + i];
#pragma unroll
    for (int j = 0; j < 16; ++j) { acc.x |= v[j].x; acc.y |= v[j].y; acc.z |= v[j].z; acc.w |= v[j].w; }
  }
  for (; k + 4u <= parts; k += 4u) {
    uint4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = first[(size_t)(k + (u32)j) * Qs + i];
#pragma unroll
    for (int j = 0; j < 4; ++j) { acc.x |= v[j].x; acc.y |= v[j].y; acc.z |= v[j].z; acc.w |= v[j].w; }
  }
  for (; k < parts; ++k) {
    const uint4 v = first[(size_t)k * Qs + i];
    acc.x |= v.x; acc.y |= v.y; acc.z |= v.z; acc.w |= v.w;
  }
  first[i] = acc;
}

constexpr int D2_NEWBITS_NT = 1024;      // threads per workgroup of k_d2_newbits: ONE add to the list's counter per workgroup and trip
// Three shapes, chosen grid-uniformly from what the level's push left behind:
//   * the push appended its discoveries to the list itself (a sparse level, bfs_fused_sparse.hpp): nothing to do;
//   * the push stored more than declare_mul x list_cap marks (slot_marks: one add per push workgroup): the list is DECLARED
//     overflowed (count = list_cap + 1: the level takes the bitmap exchange, which is always right) and the sweep is a plain
//     stream -- no barriers, no counter;
//   * otherwise: the sweep that also compacts the ids.
// The bitmaps of the cold-edge pass: ONE per slice -- k_d2_cold_reduce has ORed the slice's buffers into the first.  (This
// kernel used to OR them itself: a
// thousand 80 KB buffers on RMAT-26 / 8, one 4-byte load per lane, buffer and trip -- 75 of the sweep's 105 us on the big levels;
// eight loads in flight: 57 of 74; 16-byte loads by lane groups with a run of 1024 vertices per wave: slower, 109.)
__global__ __launch_bounds__(D2_NEWBITS_NT) void k_d2_newbits(const u32* __restrict__ visited, const unsigned char* __restrict__ mark,
                                                              u32* __restrict__ out, long long nwords, long long n, bfs_ctrl_t* c,
                                                              u32* __restrict__ list, u32 list_cap, d2_cold_view_t cv, int level,
                                                              const u32* __restrict__ slot_marks, u32 declare_mul,
                                                              const u32* __restrict__ defer_buf, u32 defer_words, int defer_reduced = 1) {
  constexpr int NT = D2_NEWBITS_NT, NW = NT / WAVE;
  __shared__ u32 s_wave[NW];
  __shared__ u32 s_base;
  if (bfs_d2_frozen(c, level)) return;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    // (what all ranks discovered in the level before: the next traversal's level plan is made from these, d2_run)
    if (level > 0 && level <= 64) c->d2_level_new[level - 1] = c->merged_new > 0xFFFFFFFFull ? 0xFFFFFFFFu : (u32)c->merged_new;
    c->merged_new = 0;                                            // the merge of this level counts into it
  }
  if (c->d2_append_level == level) {                              // (grid-uniform: written by the level's push launch)
    // (counted HERE for mgx_dbfs2_path_levels: the same add in the push launch's opener cost that kernel 17 % -- 534 -> 621 us per
    //  traversal on a rank of RMAT-26 / 8, every level slower, for one read-modify-write by one thread; found by bisecting two builds)
    if (blockIdx.x == 0 && threadIdx.x == 0) c->small_levels += 1;
    return;
  }
  if (list && slot_marks && declare_mul) {
    u64 M = 0;
#pragma unroll
    for (int i = 0; i < BFS_MARK_CTRS; ++i) M += slot_marks[((level & 1) * BFS_MARK_CTRS + i) * BFS_MARK_STRIDE];
    if (M > (u64)list_cap * (u64)declare_mul) {
      if (blockIdx.x == 0 && threadIdx.x == 0) { list[0] = list_cap + 1u; c->d2_declared_level = level; }
      list = nullptr;
    }
  }
  const bool with_cold = cv.flush && c->cold_slot == level;       // (grid-uniform) the level's push ran the cold-edge pass
  // (grid-uniform) push workgroups of the level left their hot marks as bitmaps (bfs_hot_epilogue): ORed into the first by k_d2_cold_reduce
  const bool with_defer = defer_buf && c->flush_count[level & 1] > 0u;
  const int wave = threadIdx.x / WAVE;
  const long long stride = (long long)gridDim.x * NT;
  for (long long w0 = (long long)blockIdx.x * NT; w0 < nwords; w0 += stride) {      // (block-uniform trip count: the barriers below)
    const long long w = w0 + threadIdx.x;
    u32 bits = 0;
    if (w < nwords) {
      if (w * 32 + 32 <= n) {
        const uint4* m = (const uint4*)(mark + w * 32);              // 32 marks (0/1 bytes) -> 32 bits
        const uint4 lo = m[0], hi = m[1];
        const u32 x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) bits |= (((x[i] & 0x01010101u) * 0x10204080u) >> 28) << (4 * i);
      } else {
        for (int i = 0; i < 32 && w * 32 + i < n; ++i) bits |= (mark[w * 32 + i] ? 1u : 0u) << i;
      }
      if (with_defer && w < (long long)defer_words) {
        bits |= defer_buf[w];
        // (the reduce in front did not run -- a level the plan expected to be sparse, d2_push: the few buffers there are, one by one)
        if (!defer_reduced) {
          u32 F = c->flush_count[level & 1];
          if (F > (u32)BFS_FLUSH_MAX) F = (u32)BFS_FLUSH_MAX;
          for (u32 k = 1; k < F; ++k) bits |= defer_buf[(size_t)k * BFS_FLUSH_WORDS + w];
        }
      }
      if (with_cold) {
        // the slice this word lies in (slices start on multiples of 1024 vertices: a word belongs to one slice)
        const u32 v0 = (u32)(w * 32);
        int q = -1;
        if (cv.slice_n) {
          if (v0 >= cv.lo[0]) {
            const u32 k = (v0 - cv.lo[0]) / cv.slice_n;
            if (k < 128u && cv.qof[k] != 255) q = (int)cv.qof[k];
          }
        } else {
          for (int j = 0; j < cv.slices; ++j)
            if (v0 >= cv.lo[j] && v0 - cv.lo[j] < (u32)BFS_COLD_WORDS * 32u) q = j;
        }
        if (q >= 0) {
          const u32* const base = cv.flush + ((v0 - cv.lo[q]) >> 5);
          if (cv.reduced) {
            bits |= base[(size_t)cv.wgs[q] * BFS_COLD_WORDS];
          } else {
            for (u32 k = cv.wgs[q]; k < cv.wgs[q + 1]; ++k) bits |= base[(size_t)k * BFS_COLD_WORDS];
          }
        }
      }
      bits &= ~visited[w];
      out[w] = bits;
    }
    if (list) {                                                        // (grid-uniform)
      // positions in the list: a scan over the wave, a scan of the waves' totals, ONE add to the list's counter per workgroup --
      // on a level that discovers something everywhere every wave has a count, and they all look at the counter before any of
      // them has made it overflow: per-wave adds to the one address cost this kernel 25 of its 40 us on such a level.  A list
      // that has overflowed is not read (only "more than list_cap" matters from there on): the workgroup skips its add.
      const u32 cnt = (u32)__popc(bits);
      const u32 inc = wave_inclusive_sum(cnt);
      if (lane_id() == WAVE - 1) s_wave[wave] = inc;
      __syncthreads();
      if (wave == 0) {
        const u32 v = lane_id() < NW ? s_wave[lane_id()] : 0u;
        const u32 incl = wave_inclusive_sum(v);
        if (lane_id() < NW) s_wave[lane_id()] = incl - v;
        const u32 btot = (u32)__shfl((int)incl, NW - 1, WAVE);
        if (lane_id() == 0) {
          u32 base = 0xFFFFFFFFu;                                      // (nothing to place, or the list is over already)
          if (btot && __builtin_nontemporal_load(&list[0]) <= list_cap) base = atomicAdd(&list[0], btot);
          s_base = base;
        }
      }
      __syncthreads();
      const u32 base = s_base;
      if (base != 0xFFFFFFFFu && cnt) {
        u32 at = base + s_wave[wave] + inc - cnt;
        for (u32 rest = bits; rest; rest &= rest - 1u, ++at)
          if (at < list_cap) list[D2_LIST_HEAD + at] = (u32)(w * 32) + (u32)(__ffs((int)rest) - 1);
      }
      __syncthreads();           // s_wave / s_base are reused by the next trip
    }
  }
}

// merged[w] = OR over the maps of gathered[r][w]; visited |= merged; ctrl->merged_new += popcount(merged); clear[w] = 0.
// Every rank computes the same count, so the traversal ends on all ranks together without a reduction.
// 16 bytes per lane (nwords4 = words / 4; the buffers are padded to a multiple of 4 words); map r starts
// stride4 x 16 bytes after map r - 1.
__global__ __launch_bounds__(BLOCK) void k_d2_or(const uint4* __restrict__ gathered, int maps, long long stride4,
                                                 long long nwords4, uint4* __restrict__ merged,
                                                 uint4* __restrict__ visited, bfs_ctrl_t* c, int level, uint4* clear) {
  if (bfs_d2_frozen(c, level)) return;
  // merged = the frontier of level + 1 as a bitmap over all vertices: what the unit-block body of the next push reads
  if (blockIdx.x == 0 && threadIdx.x == 0) c->fb_slot = level + 1;
  int found = 0;
  for (long long w = (long long)blockIdx.x * BLOCK + threadIdx.x; w < nwords4; w += (long long)gridDim.x * BLOCK) {
    uint4 g = make_uint4(0, 0, 0, 0);
    for (int r = 0; r < maps; ++r) {
      const uint4 x = gathered[(long long)r * stride4 + w];
      g.x |= x.x; g.y |= x.y; g.z |= x.z; g.w |= x.w;
    }
    merged[w] = g;
    // the rank's own new-bit map is consumed: all zero again for the next level (a sparse level ORs single bits into it).  On a
    // one-rank run `gathered` IS that map: this thread has read the word it clears.
    if (clear) clear[w] = make_uint4(0u, 0u, 0u, 0u);
    if (g.x | g.y | g.z | g.w) {
      uint4 v = visited[w];
      v.x |= g.x; v.y |= g.y; v.z |= g.z; v.w |= g.w;
      visited[w] = v;
    }
    found += __popc(g.x) + __popc(g.y) + __popc(g.z) + __popc(g.w);
  }
  // one add per WORKGROUP: on a level that discovers something everywhere every wave has a count, and four thousand adds to
  // one address cost the kernel 20 of its 32 us (they queue up at ~5 ns each)
  __shared__ int s_found;
  if (threadIdx.x == 0) s_found = 0;
  __syncthreads();
  found = wave_sum(found);
  if (lane_id() == 0 && found) atomicAdd(&s_found, found);
  __syncthreads();
  if (threadIdx.x == 0 && s_found) atomicAdd(&c->merged_new, (u64)s_found);
}

// out[w] = OR over the maps of maps[r][w] (the reduce step of the slice exchange: the maps are the ranks' versions of
// this rank's slice)
__global__ __launch_bounds__(BLOCK) void k_d2_or_maps(const uint4* __restrict__ maps, int nmaps, long long stride4,
                                                      long long nwords4, uint4* __restrict__ out) {
  for (long long w = (long long)blockIdx.x * BLOCK + threadIdx.x; w < nwords4; w += (long long)gridDim.x * BLOCK) {
    uint4 g = maps[w];
    for (int r = 1; r < nmaps; ++r) {
      const uint4 x = maps[(long long)r * stride4 + w];
      g.x |= x.x; g.y |= x.y; g.z |= x.z; g.w |= x.w;
    }
    out[w] = g;
  }
}

// The sparse merge.  glists: `nlists` id lists, `stride` words apart (what the all-gather delivered).  If any list
// overflowed its capacity nothing is applied (the level takes the bitmap exchange); otherwise every listed vertex is
// decided once by atomicOr on the bitmap -- the same id may come from several ranks -- and, if this rank owns it
// (v % ranks == rank), labelled and appended to the next level's queues: one block scan per queue and ONE packed cursor
// atomic per workgroup and round, as in k_bfs_build.  host_flag (pinned): [1] overflow, [2] sum of the counts (0: the level
// found nothing anywhere -- the traversal is over), then [0] = seq, which the host spins on.  The kernel does NOT reset the
// header of this rank's own list for the next level's sweep: on a one-rank run `glists` IS that list, and a workgroup that
// starts after the reset would read a count of 0 and drop its share of the ids (there is no grid-wide barrier between the
// reads and such a store) -- d2_apply_lists clears it behind the kernel instead.
template <int NT>
__global__ __launch_bounds__(NT) void k_d2_lists_apply(bfs_fused_args_t a, int level, const u32* __restrict__ glists, int nlists, u32 stride,
                                                       u32 cap, int* __restrict__ labels, int ranks, int rank, u64* host_flag, u64 seq,
                                                       u32* own_bits, int own_list, u32* own_count, int spec) {
  constexpr int NW = NT / WAVE;
  constexpr u64 CNT1 = 1ull << 40;
  constexpr u64 DEGMASK = CNT1 - 1ull;
  __shared__ u32 s_pre[66];
  __shared__ int s_over;
  __shared__ u64 s_scan[NW + 1];
  __shared__ u64 s_base[2];
  __shared__ u32 s_long_true;
  bfs_ctrl_t* const c = a.ctrl;
  if (bfs_d2_frozen(c, level)) return;
  if (threadIdx.x == 0) {
    u32 run = 0;
    int over = 0;
    for (int r = 0; r < nlists; ++r) {
      const u32 cnt = glists[(size_t)r * stride];
      s_pre[r] = run;
      if (cnt > cap) over = 1;
      run += cnt > cap ? cap : cnt;
    }
    s_pre[nlists] = run;
    s_over = over;
    // header words 1 .. 3 of every list: the level plan its rank is about to enqueue (d2_run writes them in front of level 0).
    // All equal: host_flag[3] = 1 -- every rank reads the same gathered headers, so every rank gets the same answer.
    if (blockIdx.x == 0 && host_flag) {
      int agree = 1;
      for (int r = 1; r < nlists; ++r)
        for (int k = 1; k < D2_LIST_HEAD; ++k)
          if (glists[(size_t)r * stride + k] != glists[k]) agree = 0;
      host_flag[3] = (u64)agree;
    }
  }
  __syncthreads();
  const u32 T = s_pre[nlists];
  const bool over = s_over != 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    if (level < 64) c->d2_level_kind[level] = over ? (unsigned char)2 : (unsigned char)1;
    // a speculative plan (d2_run) sent this level through the lists and one overflowed: nothing of a later level may run until
    // the host has sent this one through the bitmap exchange.  Every rank reads the same headers: all freeze at the same level.
    // (The other workgroups of this launch pass bfs_d2_frozen either way: level > level is false.)
    if (spec && over) c->d2_frozen_level = level;
    if (host_flag) {
      host_flag[1] = over ? 1ull : 0ull;
      host_flag[2] = (u64)T;
      __threadfence_system();
      __hip_atomic_store(&host_flag[0], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
  // the count of this rank's own list, for the next level: here when the gathered lists are a COPY (more than one rank: nobody in
  // this launch reads the rank's own buffer); on a one-rank run d2_apply_lists clears it behind the kernel (see the header)
  if (own_count && blockIdx.x == 0 && threadIdx.x == 0) own_count[0] = 0u;
  if (over || T == 0u) return;
  const u32 long_min = a.long_min > 0 ? (u32)a.long_min : 0xFFFFFFFFu;
  u64* const cur_s = &c->cursor[(level + 1) % 3];
  u64* const cur_l = &c->lcursor[(level + 1) % 3];
  u32* __restrict__ const out_row_s = a.fr_row[(level + 1) & 1];
  u32* __restrict__ const out_off_s = a.fr_off[(level + 1) & 1];
  u32* __restrict__ const out_row_l = a.lq_row[(level + 1) & 1];
  u32* __restrict__ const out_off_l = a.lq_off[(level + 1) & 1];
  for (u32 base = blockIdx.x * NT; base < T; base += gridDim.x * NT) {        // (block-uniform)
    const u32 idx = base + threadIdx.x;
    bool fresh = false, mine = false;
    u32 ro = 0, deg = 0;
    if (idx < T) {
      int r = 0;
      while (r + 1 < nlists && s_pre[r + 1] <= idx) ++r;
      const u32 v = glists[(size_t)r * stride + D2_LIST_HEAD + (idx - s_pre[r])];
      const u32 bit = 1u << (v & 31u);
      // the level is merged from the lists: this rank's new-bit map is not shipped -- its words go back to zero (every bit in it
      // is one of the rank's own list: stores of 0 from several threads to one word are the same store)
      if (r == own_list && own_bits) own_bits[v >> 5] = 0u;
      fresh = !(atomicOr(a.visited + (v >> 5), bit) & bit);
      mine = fresh && (int)(v % (u32)ranks) == rank;
      if (mine) {
        const u32 local = v / (u32)ranks;
        labels[local] = level + 1;
        const bfs_u32x2 ext = *(const bfs_u32x2*)(a.row_offsets + local);
        ro = ext.x; deg = ext.y - ext.x;
      }
    }
    const u32 nfresh = wave_sum(fresh ? 1u : 0u), nmine = wave_sum(mine ? 1u : 0u);
    if (lane_id() == 0) {
      if (nfresh) atomicAdd(&c->merged_new, (u64)nfresh);
      if (nmine) atomicAdd(&c->reached, (u64)nmine);
    }
    const bool is_long = mine && deg >= long_min;
    const u64 add_s = (mine && !is_long && deg) ? (CNT1 | (u64)deg) : 0ull;
    const u64 add_l = is_long ? (CNT1 | (u64)bfs_lq_pad(deg)) : 0ull;
    if (threadIdx.x == 0) s_long_true = 0;
    u64 tot_s, tot_l;
    const u64 ex_s = block_exclusive_sum_lean<NW>(add_s, s_scan, &tot_s);    // (also orders s_long_true = 0 before the adds)
    const u64 ex_l = block_exclusive_sum_lean<NW>(add_l, s_scan, &tot_l);
    const u32 lt = wave_sum(is_long ? deg : 0u);
    if (lane_id() == 0 && lt) atomicAdd(&s_long_true, lt);
    __syncthreads();
    if (threadIdx.x == 0) {
      s_base[0] = (tot_s >> 40) ? atomicAdd(cur_s, ((tot_s >> 40) << BFS_VSHIFT) | (tot_s & DEGMASK)) : 0ull;
      s_base[1] = (tot_l >> 40) ? atomicAdd(cur_l, ((tot_l >> 40) << BFS_VSHIFT) | (tot_l & DEGMASK)) : 0ull;
      if (tot_l >> 40) atomicAdd(&c->ledges[(level + 1) % 3], (u64)s_long_true);
    }
    __syncthreads();
    if (mine && deg) {
      const u64 b = is_long ? s_base[1] : s_base[0];
      const u64 at = is_long ? ex_l : ex_s;
      const u64 slot = (b >> BFS_VSHIFT) + (at >> 40);
      (is_long ? out_row_l : out_row_s)[slot] = ro;
      (is_long ? out_off_l : out_off_s)[slot] = (u32)((b & BFS_EMASK) + (at & DEGMASK)) | (is_long ? (deg & 63u) : 0u);
    }
    __syncthreads();           // s_base / s_long_true are reused by the next round
  }
}

// end of a freeze (one thread; enqueued by the host in front of the frozen level's bitmap exchange)
__global__ void k_d2_unfreeze(bfs_ctrl_t* c) {
  if (blockIdx.x == 0 && threadIdx.x == 0) c->d2_frozen_level = -1;
}

// unit owners as the builder numbers them (local rows; n_local for padding units) -> global ids (n_global for padding)
__global__ __launch_bounds__(BLOCK) void k_d2_owner_global(int* __restrict__ owner, long long units_pad, int ranks, int rank, int n_local,
                                                           int n_global) {
  const long long i = (long long)blockIdx.x * BLOCK + threadIdx.x;
  if (i >= units_pad) return;
  const int o = owner[i];
  owner[i] = (o >= n_local) ? n_global : o * ranks + rank;
}

// rows of at least min_deg entries (a prefix of the rank's rows: they come by descending degree) and whether every row is
// sorted by neighbour id (the cold-list builder finds a row's cold tail by bisection): *long_rows <- the count, *sorted <- 0 if not
__global__ __launch_bounds__(BLOCK) void k_d2_row_facts(const int* __restrict__ ro, const int* __restrict__ ci, int n_local, int min_deg,
                                                        int* long_rows, int* sorted) {
  const long long nwaves = ((long long)gridDim.x * BLOCK) / WAVE;
  bool bad = false;
  for (long long r = ((long long)blockIdx.x * BLOCK + threadIdx.x) / WAVE; r < n_local; r += nwaves) {
    const int r0 = ro[r], r1 = ro[r + 1];
    if (lane_id() == 0 && r1 - r0 >= min_deg) atomicMax(long_rows, (int)r + 1);
    for (int e = r0 + 1 + lane_id(); e < r1; e += WAVE) bad |= ci[e - 1] > ci[e];
  }
  if (__ballot(bad) && lane_id() == 0) *sorted = 0;
}

// Start of a traversal: the rank's labels (-1), visited bitmap, mark bytes, new-bit map and list header cleared by ONE launch (they
// were five fills of 4 B .. 64 MB, each a launch of its own: ~45 us per traversal on a rank of RMAT-26 / 8).  Regions start on
// 16-byte boundaries (device allocations); a region's last bytes, if it is not a multiple of 16 long, are written one by one.
struct d2_fill_t { void* p; size_t bytes; u32 word; };
__global__ __launch_bounds__(BLOCK) void k_d2_clear(d2_fill_t r0, d2_fill_t r1, d2_fill_t r2, d2_fill_t r3, d2_fill_t r4) {
  const d2_fill_t regs[5] = {r0, r1, r2, r3, r4};
  const size_t tid = (size_t)blockIdx.x * BLOCK + threadIdx.x, nth = (size_t)gridDim.x * BLOCK;
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const d2_fill_t r = regs[k];
    if (!r.p || !r.bytes) continue;
    const uint4 v = make_uint4(r.word, r.word, r.word, r.word);
    uint4* const q = (uint4*)r.p;
    const size_t n16 = r.bytes / 16;
    for (size_t i = tid; i < n16; i += nth) q[i] = v;
    if (tid == 0)
      for (size_t b = n16 * 16; b < r.bytes; ++b) ((unsigned char*)r.p)[b] = (unsigned char)(r.word & 0xFFu);
  }
}

// list / plan_levels / plan_sparse: the level plan this rank is about to enqueue goes into header words 1 .. 3 of its id list (d2_run's
// agreement; NULL: no list)
__global__ void k_d2_init(bfs_fused_args_t a, int* labels_local, int src, int ranks, int rank, u32* list = nullptr, u32 plan_levels = 0,
                          u64 plan_sparse = 0) {
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  if (list) { list[1] = plan_levels; list[2] = (u32)plan_sparse; list[3] = (u32)(plan_sparse >> 32); }
  bfs_ctrl_reset(a.ctrl);
  bfs_slot_marks_clear(a, 0);
  bfs_slot_marks_clear(a, 1);
  a.ctrl->fb_slot = -1;                        // (the merged bitmap describes a level only once k_d2_or has written it)
  a.visited[src >> 5] = 1u << (src & 31);      // every rank knows the source is visited
  if (src % ranks == rank) {
    labels_local[src / ranks] = 0;
    bfs_seed_queue(a, (u32)(src / ranks));
  }
}

struct d2_state_t {
  int n_global = 0, n_local = 0, ranks = 1, rank = 0;
  const int* row_offsets = nullptr;   // local rows (n_local + 1)
  const int* col_indices = nullptr;   // global hub-first ids
  std::unique_ptr<bfs_fused_state_t> fs;
  mem_t<int> labels;                  // n_local
  mem_t<u32> merged;                  // n_global bits
  u32* newbits = nullptr;             // caller-owned (n_global bits): what this rank discovered in the level
  long long nwords = 0;
  long long last_edges = 0;           // edges of the frontier the last push expanded
  int cold_forced = -1;               // MGX_BFS_COLD_TEST at creation (-1: by size)
  // id lists of the sparse levels: the rank's own (caller-owned or ours), its capacity in ids; 0: bitmaps on every level
  u32* mylist = nullptr;
  u32 list_cap = 0;
  mem_t<u32> own_list;
  mem_t<u32> slot_marks;              // marks stored by the push workgroups of a level (bfs_body_finish): what k_d2_newbits bases its shape on
  u32 declare_mul = 16;               // a level whose push stored more than this x list_cap marks does not fill its list (MGX_DIST_DECLARE_MUL; 0: always fill)
  int sparse_push = 1;                // levels of at most list_cap edges append to the list themselves (MGX_DIST_SPARSE_PUSH=0: every level sweeps)
  u64* host_flag = nullptr;           // pinned: what k_d2_lists_apply tells the host
  u64 flag_seq = 0;
  // unit blocks of the rank's long rows (mgx_layout.hip; owners in GLOBAL ids, padding units owned by vertex n_global):
  // built on request (mgx_dbfs2_build_units), owned here
  int* ub_col = nullptr;
  int* ub_owner = nullptr;
  // the short rows vertex by vertex (bfs_fused_vshort.hpp; mgx_dbfs2_build_units sets it up when the rank's rows come by
  // non-increasing degree): class boundaries over the LOCAL rows, the edges of those rows, a copy of col_indices with readable
  // entries behind it, the frontier over the local rows (written by k_bfs_build2)
  u32 vs_v[4] = {0, 0, 0, 0};
  u32 vs_v9 = 0, vs_edges = 0, vs_div = 0;
  mem_t<int> col_pad;
  mem_t<u32> front_local;
  mem_t<u32> ub_col24;                // the same, 24 bits per entry: only when the blocks hold the rows' HOT entries alone (the others are in the cold-edge lists)
  long long ub_units = 0, ub_units_pad = 0;
  u32 dense_div = 4;
  // cold-edge lists of those rows (mgx_layout.hip: mgx_cold_build_device; owners global): pairs by slice of the destination,
  // the slices that hold any, the cold workgroups of a push launch per slice, their flush bitmaps
  int* cold_owner = nullptr;
  int* cold_dst = nullptr;
  u32* cold_pk = nullptr;             // the same pairs at four bytes each (mgx_layout.hip: mgx_cold_pack_device), their 64-chunks' owners
  u32* cold_cbase = nullptr;
  u32 cold_cb[BFS_COLD_MAX_SLICES + 1] = {};
  u64 cold_pk_mask = 0;
  long long cold_pairs = 0;
  int cold_slices = 0;
  u32 cold_lo[BFS_COLD_MAX_SLICES] = {};
  u32 cold_off[BFS_COLD_MAX_SLICES + 1] = {};
  u32 cold_wgs[BFS_COLD_MAX_SLICES + 1] = {};
  mem_t<u32> cold_flush;
  int cold_reduce = 1;                // k_d2_cold_reduce in front of the sweep (MGX_DIST_COLD_REDUCE)
  int fused_merge = 1;                // OR-merge inside the queue build (MGX_DIST_FUSED_MERGE)
  int build_list = 0;                 // the list-based queue build (MGX_DIST_BUILD_LIST)
  int push_split = 0;                 // measurements: the push grid's three parts as three launches (MGX_DIST_PUSH_SPLIT)
  u32 grid_div = 1;                   // the push grid's long-row and short-row halves get 2 x CUs / grid_div workgroups each (set by the shard builder: a small shard takes fewer, fatter workgroups)
  bool skip_small_reduce = false;     // d2_run's plan: no k_d2_cold_reduce launch on the levels it expects to be merged from id lists
  mem_t<u32> defer_buf;               // deferred hot marks of the push workgroups (bfs_hot_epilogue): BFS_FLUSH_MAX bitmaps; empty: nothing is deferred (MGX_DIST_DEFER=0)
  // ... and only on a shard big enough to pay for the 80 KB bitmap every deferring workgroup writes and the reduce behind it:
  // RMAT-25 / 8 (134 M entries per rank) 551 against 578 us of kernels per traversal with them, RMAT-22 / 8 (17 M) 264 against 225
  // without.  mgx_dbfs2_build_units sets it from the rank's entry count (MGX_DIST_DEFER=2: always)
  bool defer_pays = true;
  static constexpr long long D2_DEFER_MIN_ENTRIES = 48ll << 20;
  d2_cold_view_t cold_view() const {
    d2_cold_view_t v;
    if (!cold_dst || cold_slices <= 0 || !cold_flush.size()) return v;
    v.flush = cold_flush.data(); v.slices = cold_slices;
    // (the table covers 128 slices behind the first one in use -- 83 M vertices; a bigger range keeps the search over lo[])
    v.slice_n = (u32)BFS_COLD_WORDS * 32u;
    bool table = true;
    for (int k = 0; k < 128; ++k) v.qof[k] = 255;
    for (int q = 0; q < cold_slices; ++q) {
      const u32 k = (cold_lo[q] - cold_lo[0]) / v.slice_n;
      if (k >= 128u || (cold_lo[q] - cold_lo[0]) % v.slice_n) { table = false; break; }
      v.qof[k] = (unsigned char)q;
    }
    if (!table) v.slice_n = 0;
    v.reduced = cold_reduce ? 1 : 0;
    for (int i = 0; i < BFS_COLD_MAX_SLICES; ++i) v.lo[i] = cold_lo[i];
    for (int i = 0; i <= BFS_COLD_MAX_SLICES; ++i) v.wgs[i] = cold_wgs[i];
    return v;
  }

  void init(standard_context_t& ctx, int n_global_, int ranks_, int rank_, const int* ro, const int* ci, u32* newbits_) {
    n_global = n_global_; ranks = ranks_; rank = rank_;
    n_local = (n_global - rank + ranks - 1) / ranks;
    row_offsets = ro; col_indices = ci; newbits = newbits_;
    nwords = (((long long)n_global + 31) / 32 + 3) / 4 * 4;     // padded to 16 bytes: the OR-merge reads uint4
    fs.reset(new bfs_fused_state_t(n_global, ctx));
    // (a rank's short rows take the queue search, not the vertex-by-vertex walk the single-GPU threshold was re-tuned for in
    //  round 4: the ranks keep 64 unless the switch says otherwise)
    if (!mgx::env("MGX_BFS_LONG_MIN")) fs->long_min = 64;
    if (const char* e = mgx::env("MGX_BFS_COLD_TEST")) cold_forced = atoi(e);
    if (const char* e = mgx::env("MGX_DIST_DECLARE_MUL")) declare_mul = (u32)atoi(e);
    if (const char* e = mgx::env("MGX_DIST_SPARSE_PUSH")) sparse_push = atoi(e);
    if (const char* e = mgx::env("MGX_DIST_FUSED_MERGE")) fused_merge = atoi(e);
    if (const char* e = mgx::env("MGX_DIST_BUILD_LIST")) build_list = atoi(e);
    if (const char* e = mgx::env("MGX_DIST_PUSH_SPLIT")) push_split = atoi(e);
    {
      int defer = 1;
      if (const char* e = mgx::env("MGX_DIST_DEFER")) defer = atoi(e);
      if (defer) defer_buf = mem_t<u32>((size_t)BFS_FLUSH_MAX * BFS_FLUSH_WORDS, ctx);
    }
    slot_marks = mem_t<u32>((size_t)2 * BFS_MARK_CTRS * BFS_MARK_STRIDE, ctx);
    MGX_HIP(hipMemsetAsync(slot_marks.data(), 0, slot_marks.size() * sizeof(u32), ctx.stream()));
    labels = mem_t<int>((size_t)n_local + 1, ctx);
    merged = mem_t<u32>((size_t)nwords + 4, ctx);
    MGX_HIP(hipMemsetAsync(merged.data(), 0, ((size_t)nwords + 4) * sizeof(u32), ctx.stream()));   // (the bit of vertex n_global stays 0)
    MGX_HIP(hipHostMalloc((void**)&host_flag, 64, hipHostMallocDefault));
    host_flag[0] = host_flag[1] = host_flag[2] = 0;
  }
  ~d2_state_t() {
    if (host_flag) (void)hipHostFree(host_flag);
    if (ub_col) (void)hipFree(ub_col);
    if (ub_owner) (void)hipFree(ub_owner);
    if (cold_owner) (void)hipFree(cold_owner);
    if (cold_dst) (void)hipFree(cold_dst);
    if (cold_pk) (void)hipFree(cold_pk);
    if (cold_cbase) (void)hipFree(cold_cbase);
  }
  d2_state_t() {}
  d2_state_t(const d2_state_t&) = delete;
  d2_state_t& operator=(const d2_state_t&) = delete;
  // capacity (ids) of a rank's list: n / (256 ranks), at least 252, so that head + ids is a multiple of 4 words
  static u32 default_list_cap(int n_global, int ranks) {
    long long c = (long long)n_global / (256ll * ranks);
    if (c < 252) c = 252;
    c = (c + D2_LIST_HEAD + 3) / 4 * 4 - D2_LIST_HEAD;
    return (u32)c;
  }
  void set_list(u32* d_list, u32 cap) { mylist = d_list; list_cap = d_list ? cap : 0u; }
  long long list_words() const { return (long long)D2_LIST_HEAD + list_cap; }
  bfs_fused_args_t args() const {
    bfs_fused_args_t a{};
    a.row_offsets = (const u32*)row_offsets;
    a.col_indices = col_indices;
    a.labels = nullptr;
    a.visited = fs->visited.data();
    a.mark = fs->mark.data();
    // (with unit blocks: the level's merged discoveries ARE the frontier bitmap their owners are looked up in)
    a.frontier_bits = ub_col ? merged.data() : fs->frontier_bits.data();
    a.in_offsets = nullptr; a.in_indices = nullptr;
    for (int i = 0; i < 2; ++i) {
      a.fr_row[i] = fs->fr_row[i].data(); a.fr_off[i] = fs->fr_off[i].data();
      a.lq_row[i] = fs->lq_row[i].data(); a.lq_off[i] = fs->lq_off[i].data();
    }
    a.long_min = fs->long_min;
    a.hot_min_edges = fs->hot_min_edges;
    a.ctrl = fs->ctrl.data();
    a.old_of_new = nullptr; a.new_of_old = nullptr;
    a.n = n_global;
    a.mode = 0; a.alpha = 0.f;
    a.count_marks = 0;
    a.ub_col = ub_col; a.ub_col24 = ub_col24.size() ? ub_col24.data() : nullptr; a.ub_owner = ub_owner; a.ub_units = (u32)ub_units; a.ub_units_pad = (u32)ub_units_pad; a.dense_div = ub_col ? dense_div : 0u;
    {
      const bool vs = vs_div != 0u && col_pad.size() && front_local.size() && !build_list;      // (the list-based build writes no local frontier)
      for (int i = 0; i < 4; ++i) a.vs_v[i] = vs ? vs_v[i] : 0u;
      a.vs_v9 = vs ? vs_v9 : 0u; a.vs_edges = vs ? vs_edges : 0u; a.vs_div = vs ? vs_div : 0u; a.vs_dummy = 0;     // (entry 0: readable, and a lane without entries looks at none of the four)
      a.vs_col = vs ? col_pad.data() : nullptr; a.d2_front = vs ? const_cast<u32*>(front_local.data()) : nullptr;
    }
    a.lazy_div = 0; a.slot_marks = const_cast<u32*>(slot_marks.data()); a.merged_pull = 0; a.lazy_pull = 0; a.chain_big_edges = 0; a.defer_reach_mul = 1; a.defer_reach_div = 1;
    const bool cold = cold_dst != nullptr && ub_col != nullptr && cold_slices > 0 && cold_flush.size() > 0;
    a.cold_owner = cold ? cold_owner : nullptr; a.cold_dst = cold ? cold_dst : nullptr; a.cold_slices = cold ? cold_slices : 0;
    a.cold_flush = cold ? const_cast<u32*>(cold_flush.data()) : nullptr;
    const bool pk = cold && cold_pk && cold_cbase;
    a.cold_pk = pk ? cold_pk : nullptr; a.cold_cbase = pk ? cold_cbase : nullptr; a.cold_pk_mask = pk ? cold_pk_mask : 0ull; a.cold_ranks = (u32)ranks;
    for (int i = 0; i <= BFS_COLD_MAX_SLICES; ++i) a.cold_cb[i] = pk ? cold_cb[i] : 0u;
    for (int i = 0; i < BFS_COLD_MAX_SLICES; ++i) a.cold_lo[i] = cold ? cold_lo[i] : 0u;
    for (int i = 0; i <= BFS_COLD_MAX_SLICES; ++i) { a.cold_off[i] = cold ? cold_off[i] : 0u; a.cold_wgs[i] = cold ? cold_wgs[i] : 0u; }
    // deferred hot marks: while the ranks together have reached fewer vertices than the deferred range holds (reached counts this
    // rank's: x ranks); k_d2_cold_reduce + k_d2_newbits read the bitmaps
    const bool defer = defer_buf.size() != 0 && cold_reduce && defer_pays;
    a.flush_buf = defer ? const_cast<u32*>(defer_buf.data()) : nullptr; a.defer_min_marks = defer ? fs->defer_min_marks : 0u;
    a.defer_words = defer ? (u32)BFS_FLUSH_WORDS : 0u; a.defer_reach_mul = (u32)ranks; a.defer_reach_div = 1;
    a.chain_max_edges = 0;               // (levels are counted by the host here: no chains of small levels)
    const bool sparse = sparse_push && mylist && list_cap > 0u;
    a.d2_list = sparse ? mylist : nullptr; a.d2_list_cap = sparse ? list_cap : 0u; a.d2_newbits = sparse ? newbits : nullptr;
    return a;
  }
};

// Start of a traversal (asynchronous).
inline void d2_reset(d2_state_t& st, int src, standard_context_t& ctx, u32 plan_levels = 0, u64 plan_sparse = 0) {
  hipStream_t s = ctx.stream();
  // (the new-bit map is all zero between the levels of a traversal that ran to its end; one that was abandoned may have left bits)
  const bool bits = st.mylist && st.sparse_push;
  const d2_fill_t labels{st.labels.data(), (size_t)st.n_local * sizeof(int), 0xFFFFFFFFu};
  const d2_fill_t visited{st.fs->visited.data(), st.fs->visited.size() * sizeof(u32), 0u};
  const d2_fill_t marks{st.fs->mark.data(), st.fs->mark.size(), 0u};
  const d2_fill_t head{st.mylist, st.mylist ? D2_LIST_HEAD * sizeof(u32) : 0, 0u};
  const d2_fill_t newbits{bits ? st.newbits : nullptr, bits ? (size_t)st.nwords * sizeof(u32) : 0, 0u};
  if (((uintptr_t)labels.p | (uintptr_t)visited.p | (uintptr_t)marks.p | (uintptr_t)head.p | (uintptr_t)newbits.p) % 16 == 0) {
    hipLaunchKernelGGL(k_d2_clear, dim3(ctx.num_cus * 8), dim3(BLOCK), 0, s, labels, visited, marks, head, newbits);
  } else {                                         // (a caller's buffer off the 16-byte grid: the fills one by one)
    MGX_HIP(hipMemsetAsync(labels.p, 0xFF, labels.bytes, s));
    MGX_HIP(hipMemsetAsync(visited.p, 0, visited.bytes, s));
    MGX_HIP(hipMemsetAsync(marks.p, 0, marks.bytes, s));
    if (head.p) MGX_HIP(hipMemsetAsync(head.p, 0, head.bytes, s));
    if (newbits.p) MGX_HIP(hipMemsetAsync(newbits.p, 0, newbits.bytes, s));
  }
  hipLaunchKernelGGL(k_d2_init, dim3(1), dim3(64), 0, s, st.args(), st.labels.data(), src, st.ranks, st.rank, st.mylist, plan_levels, plan_sparse);
}

// level kernels on the local queues (marks), then new_bits = marks & ~bitmap
inline void d2_push(d2_state_t& st, int level, standard_context_t& ctx, bool want_list = true) {
  hipStream_t s = ctx.stream();
  bfs_fused_args_t a = st.args();
  bfs_set_kernel_attributes();
  if (st.push_split) {     // measurements: cold pass, long rows, short rows as three launches
    bfs_launch_push(a, level, ctx, 2 | ((2 | 4) << 4), bfs_cold_test(a.n, st.cold_forced));
    bfs_launch_push(a, level, ctx, 0 | ((1 | 4) << 4), bfs_cold_test(a.n, st.cold_forced));
    bfs_launch_push(a, level, ctx, 0 | ((1 | 2) << 4), bfs_cold_test(a.n, st.cold_forced));
  } else
  bfs_launch_push(a, level, ctx, 2, bfs_cold_test(a.n, st.cold_forced), st.grid_div);   // (the level's bookkeeping rides on the push launch)
  d2_cold_view_t cv = st.cold_view();
  u32* const dbuf = a.flush_buf;
  // The stream reduce of the cold pass's bitmaps in front of the sweep -- not on a level the plan expects to be sparse (want_list:
  // its frontier is too small for the cold pass to run, and a launch that finds nothing to do still costs ~2.5 us; should the pass
  // have run after all, the sweep ORs the slices' buffers itself, as it did before the reduce existed).  Deferred hot marks need
  // the reduce whatever the level (the sweep reads their first buffer only).
  // (round 6: with deferred hot marks too -- the sweep then ORs the level's flush buffers itself, a handful on a level that is sparse;
  //  an empty launch is ~4.5 us, five sparse levels per traversal)
  const bool planned_small = want_list && st.skip_small_reduce;
  if (planned_small) cv.reduced = 0;
  if (((cv.flush && cv.reduced) || dbuf) && !planned_small) {
    d2_cold_view_t rv = cv;
    if (!(cv.flush && cv.reduced)) { rv.flush = nullptr; rv.slices = 0; }
    const size_t threads = (size_t)rv.slices * (BFS_COLD_WORDS / 4) + (dbuf ? (size_t)BFS_FLUSH_WORDS / 4 : 0);
    hipLaunchKernelGGL(k_d2_cold_reduce, dim3((unsigned)((threads + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, s, rv,
                       (const bfs_ctrl_t*)a.ctrl, level, rv.flush ? const_cast<u32*>(st.cold_flush.data()) : nullptr, dbuf);
  }
  hipLaunchKernelGGL(k_d2_newbits, dim3(grid_for(st.nwords, D2_NEWBITS_NT, ctx.num_cus * 2)), dim3(D2_NEWBITS_NT), 0, s, st.fs->visited.data(),
                     st.fs->mark.data(), st.newbits, st.nwords, (long long)st.n_global, a.ctrl, want_list ? st.mylist : (u32*)nullptr, st.list_cap, cv, level,
                     (const u32*)st.slot_marks.data(), st.declare_mul, (const u32*)dbuf, a.defer_words, planned_small ? 0 : 1);
}

// The sparse merge of a level (k_d2_lists_apply) on `nlists` gathered lists, `stride_words` apart, and the host's wait for
// its verdict: out3 = { 1 if some list overflowed (nothing was applied: exchange the bitmaps), sum of the counts (0: the
// level found nothing on any rank), 0 }.  Synchronises with the kernel's first workgroup only (a spin on pinned memory).
// spec: the level is part of a speculative plan -- nobody waits for the verdict; an overflow freezes the traversal (k_d2_lists_apply)
inline u64 d2_enqueue_apply_lists(d2_state_t& st, int level, const u32* glists, int nlists, long long stride_words, standard_context_t& ctx,
                                  bool spec) {
  hipStream_t s = ctx.stream();
  bfs_fused_args_t a = st.args();
  const u64 seq = spec ? 0ull : ++st.flag_seq;
  // (does the kernel read the rank's own buffer?  a one-rank run, or an in-place gather)
  const bool own_is_read = st.mylist && glists < st.mylist + st.list_words() && st.mylist < glists + (size_t)nlists * (size_t)stride_words;
  hipLaunchKernelGGL(k_d2_lists_apply<BLOCK>, dim3(256), dim3(BLOCK), 0, s, a, level, glists, nlists, (u32)stride_words, st.list_cap,
                     st.labels.data(), st.ranks, st.rank, spec ? (u64*)nullptr : st.host_flag, seq, st.newbits, nlists == 1 ? 0 : st.rank,
                     (st.mylist && !own_is_read) ? st.mylist : nullptr, spec ? 1 : 0);
  MGX_CHECK_LAUNCH("partitioned BFS: list merge launch");
  // the count of this rank's own list, for the next level's sweep: behind the kernel when the kernel reads that very list (see its header)
  if (st.mylist && own_is_read) MGX_HIP(hipMemsetAsync(st.mylist, 0, sizeof(u32), s));
  return seq;
}
inline void d2_apply_lists(d2_state_t& st, int level, const u32* glists, int nlists, long long stride_words, standard_context_t& ctx,
                           long long* out3) {
  hipStream_t s = ctx.stream();
  const u64 seq = d2_enqueue_apply_lists(st, level, glists, nlists, stride_words, ctx, false);
  volatile u64* const flag = st.host_flag;
  long long spins = 0;
  while (flag[0] != seq) {
    if (++spins > 20000000LL) { MGX_HIP(hipStreamSynchronize(s)); break; }
    __builtin_ia32_pause();
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  out3[0] = (long long)flag[1];
  out3[1] = (long long)flag[2];
  out3[2] = (long long)flag[3];        // (level 0: every rank announced the same level plan)
}

// gathered: `maps` new-bit maps, `stride_words` apart (a multiple of 4): every rank's map after an all-gather, or
// ONE map that is already the OR of all ranks' (reduce-scatter + all-gather by the caller).  Asynchronous: OR-merge
// and queue build are enqueued on the context's stream, nothing is read back.
inline void d2_merge(d2_state_t& st, int level, const u32* gathered, int maps, long long stride_words,
                     standard_context_t& ctx) {
  hipStream_t s = ctx.stream();
  bfs_fused_args_t a = st.args();
  // OR-merge and queue build in ONE launch (k_bfs_build2<., 2, RANKS>) when the ranks are a power of two (MGX_DIST_FUSED_MERGE=0:
  // the two launches below)
  const bool build_list = st.build_list != 0, fused = st.fused_merge != 0;
  const int R = st.ranks;
  if (fused && !build_list && (R == 2 || R == 4 || R == 8 || R == 16) && (uintptr_t)a.row_offsets % 16 == 0 &&
      (uintptr_t)gathered % 16 == 0 && stride_words % 4 == 0) {
    bfs_d2_fuse_t fz;
    fz.maps = gathered; fz.nmaps = maps; fz.stride = stride_words; fz.nwords = st.nwords;
    fz.merged = st.merged.data(); fz.clear = st.newbits; fz.list_head = st.mylist;
    const long long rw = R / 2;
    long long groups = ((long long)st.n_local + 15) / 16;
    if ((st.nwords + rw - 1) / rw > groups) groups = (st.nwords + rw - 1) / rw;      // (the grid covers the bitmap's words, not only the rank's rows)
    const dim3 grid(bfs_build_grid(groups * 16, 512));
#define MGX_D2_FUSED(R_) hipLaunchKernelGGL((k_bfs_build2<512, 2, R_>), grid, dim3(512), 0, s, a, level, st.labels.data(), st.n_local, \
                                            (const u32*)nullptr, st.ranks, st.rank, fz)
    if (R == 2) MGX_D2_FUSED(2); else if (R == 4) MGX_D2_FUSED(4); else if (R == 8) MGX_D2_FUSED(8); else MGX_D2_FUSED(16);
#undef MGX_D2_FUSED
    return;
  }
  if (st.mylist) MGX_HIP(hipMemsetAsync(st.mylist, 0, sizeof(u32), s));      // (the list's count, as the fused merge resets it)
  hipLaunchKernelGGL(k_d2_or, dim3(grid_for(st.nwords / 4, BLOCK, 1024)), dim3(BLOCK), 0, s, (const uint4*)gathered, maps,
                     stride_words / 4, st.nwords / 4, (uint4*)st.merged.data(), (uint4*)st.fs->visited.data(), a.ctrl, level, (uint4*)st.newbits);
  // the queue build without a list when the rank's row offsets allow its 16-byte loads (k_bfs_build2<., DIST>: labels and
  // row extents of a thread's 16 local vertices are contiguous); MGX_DIST_BUILD_LIST=1: the list-based one
  if (!build_list && ((uintptr_t)a.row_offsets % 16 == 0))
    hipLaunchKernelGGL((k_bfs_build2<512, true>), dim3(bfs_build_grid(st.n_local, 512)), dim3(512), 0, s, a, level, st.labels.data(), st.n_local,
                       (const u32*)st.merged.data(), st.ranks, st.rank);
  else
    hipLaunchKernelGGL((k_bfs_build<512, false>), dim3(bfs_build_grid(st.n_local, 512)), dim3(512), 0, s, a, level,
                       (const u32*)st.merged.data(), st.labels.data(), st.n_local, st.ranks, st.rank, 0);
}

// Synchronises and reports: out[0] traversal over (a level discovered nothing on any rank), [1] levels that hold
// vertices, [2] edges this rank expanded so far, [3] vertices of the level merged last (all ranks), [4] size and
// [5] edges of this rank's next queues.
inline void d2_status(d2_state_t& st, int next_level, standard_context_t& ctx, long long* out) {
  hipStream_t s = ctx.stream();
  bfs_ctrl_t* hc = st.fs->host_ctrl;
  MGX_HIP(hipMemcpyAsync(hc, st.fs->ctrl.data(), offsetof(bfs_ctrl_t, trace), hipMemcpyDeviceToHost, s));
  MGX_HIP(hipStreamSynchronize(s));
  const u64 cur = hc->cursor[next_level % 3], lcur = hc->lcursor[next_level % 3];
  // the level merged last may itself be the empty one: the device notices at the next level's opening, the host here
  const bool over = hc->dist_done || (next_level > 0 && hc->merged_new == 0);
  out[0] = over ? 1 : 0;
  out[1] = hc->dist_done ? hc->dist_levels : next_level;
  out[2] = (long long)hc->sum_edges;
  out[3] = (long long)hc->merged_new;
  out[4] = (long long)((cur >> BFS_VSHIFT) + (lcur >> BFS_VSHIFT));
  out[5] = (long long)((cur & BFS_EMASK) + hc->ledges[next_level % 3]);
}

// ---- the whole traversal from C++: push -> RCCL exchange -> merge for a batch of levels, one synchronisation per batch ----
// exchange 0 ("gather"): ncclAllGather of the ranks' new-bit maps, every rank ORs them (k_d2_or inside d2_merge);
// exchange 1 ("reduce"): slice r of every map goes to rank r (grouped ncclSend / ncclRecv over the xGMI full mesh: all
// seven links of a GPU busy at once), the owner ORs the slices, ncclAllGather of the merged slices: 2 (R - 1) / R bitmaps
// per rank and level instead of R - 1.  xwords: length of the exchanged map (the bitmap padded to a multiple of 4 * ranks
// words; st.newbits is that long).  Returns the status of d2_status.
struct d2_run_bufs_t {
  mem_t<u32> gathered;     // ranks * xwords (gather) / xwords (reduce: the merged map)
  mem_t<u32> recv;         // reduce: the ranks' versions of this rank's slice
  mem_t<u32> glists;       // ranks id lists (sparse levels)
  long long xwords = 0;
  int levels_hint = 8;
  // The SPECULATIVE LEVEL PLAN of a traversal with id lists (d2_run): what the last traversals on this engine looked like, level
  // by level -- how many levels, and which of them could have been merged from id lists (they were, or all ranks together
  // discovered no more than ONE list holds).  Every rank derives the same history: the verdicts come from all-gathered headers
  // and from counts every rank computes alike.
  struct hist_t { int levels = 0; u64 sparse_ok = 0; };
  hist_t hist[4];
  int hist_n = 0;
  int spec = 1;            // MGX_DIST_SPEC=0: one host look per level, as before round 5
  // traversals of the engine (statistics): planned ahead, frozen by a list that overflowed against the plan, continued level by level
  long long spec_runs = 0, spec_frozen = 0, spec_short = 0;
  int last_plan_levels = 0;
  u64 last_plan_sparse = 0;
  d2_run_bufs_t() { if (const char* e = mgx::env("MGX_DIST_SPEC")) spec = atoi(e); }
  void learn(const bfs_ctrl_t* hc, int levels, long long last_new, u32 list_cap) {
    hist_t h;
    h.levels = levels < 64 ? levels : 64;
    for (int l = 0; l < h.levels; ++l) {
      const int kind = hc->d2_level_kind[l];
      // (the level merged last has no successor whose sweep recorded its count: the merge's own counter holds it)
      const long long found = (l + 1 < levels || l >= 63) ? (long long)hc->d2_level_new[l] : last_new;
      if (kind == 1 || (kind != 2 && found <= (long long)list_cap)) h.sparse_ok |= 1ull << l;
    }
    hist[hist_n & 3] = h;
    ++hist_n;
  }
  // the plan: as many levels as the longest of the remembered traversals + the one that finds nothing; level l from id lists iff
  // every remembered traversal that had a level l could have merged it from lists (levels nobody had: lists)
  bool plan(int* levels, u64* sparse) const {
    if (!spec || hist_n == 0) return false;
    int L = 0;
    u64 sp = ~0ull;
    for (int i = 0; i < 4 && i < hist_n; ++i) {
      const hist_t& h = hist[i];
      if (h.levels > L) L = h.levels;
      const u64 have = h.levels >= 64 ? ~0ull : ((1ull << h.levels) - 1ull);
      sp &= ~have | h.sparse_ok;
    }
    *levels = L + 1;
    *sparse = sp | 1ull;          // (level 0 -- the source -- is merged from the lists in every protocol: d2_run's agreement rides on it)
    return true;
  }
};

// the bitmap exchange of one level + the dense merge (what every level did before the id lists)
inline void d2_exchange_bitmaps(d2_state_t& st, comm_t& cm, d2_run_bufs_t& bufs, int level, int exchange, long long xwords,
                                standard_context_t& ctx) {
  const rccl_api_t& api = cm.table();
  hipStream_t s = ctx.stream();
  const int R = st.ranks;
  const long long S = xwords / R;                 // words per slice (xwords is a multiple of 4 * R)
  if (R == 1 && !cm.comm) {
    d2_merge(st, level, st.newbits, 1, xwords, ctx);
  } else if (exchange == 0) {
    MGX_RCCL(api.AllGather(st.newbits, bufs.gathered.data(), (size_t)xwords, ncclUint32, cm.comm, s));
    d2_merge(st, level, bufs.gathered.data(), R, xwords, ctx);
  } else {
    {
      rccl_group_t group(api);                  // (GroupEnd on every way out: a throw inside must not leave the thread's group open)
      for (int r = 0; r < R; ++r) {
        MGX_RCCL(api.Send(st.newbits + (size_t)r * S, (size_t)S, ncclUint32, r, cm.comm, s));
        MGX_RCCL(api.Recv(bufs.recv.data() + (size_t)r * S, (size_t)S, ncclUint32, r, cm.comm, s));
      }
      group.end();
    }
    hipLaunchKernelGGL(k_d2_or_maps, dim3(grid_for(S / 4, BLOCK, 1024)), dim3(BLOCK), 0, s, (const uint4*)bufs.recv.data(), R,
                       S / 4, S / 4, (uint4*)bufs.recv.data());
    MGX_RCCL(api.AllGather(bufs.recv.data(), bufs.gathered.data(), (size_t)S, ncclUint32, cm.comm, s));
    d2_merge(st, level, bufs.gathered.data(), 1, xwords, ctx);
  }
}

inline void d2_run(d2_state_t& st, comm_t& cm, d2_run_bufs_t& bufs, int src, int exchange, long long xwords,
                   standard_context_t& ctx, long long* out6) {
  const rccl_api_t& api = cm.table();
  hipStream_t s = ctx.stream();
  const int R = st.ranks;
  if (bufs.xwords != xwords || !bufs.gathered.size()) {
    ctx.synchronize();
    bufs.gathered = mem_t<u32>((size_t)R * (size_t)xwords + 4, ctx);
    bufs.recv = mem_t<u32>((size_t)xwords + 4, ctx);
    bufs.xwords = xwords;
  }
  if (st.mylist && bufs.glists.size() < (size_t)R * (size_t)st.list_words()) {
    ctx.synchronize();
    bufs.glists = mem_t<u32>((size_t)R * (size_t)st.list_words() + 4, ctx);
  }
  // (the level plan: made BEFORE the reset, whose last launch writes it into the header of the rank's id list)
  int plan_levels = 0;
  u64 plan_sparse = 0;
  const bool have_plan = st.mylist && bufs.plan(&plan_levels, &plan_sparse);
  d2_reset(st, src, ctx, have_plan ? (u32)plan_levels : 0u, have_plan ? plan_sparse : 0ull);
  int level = 0;
  if (st.mylist) {
    // the lists of a level, all-gathered (a one-rank run without a communicator reads its own)
    auto gather_lists = [&](const u32** lists, int* nl) {
      *lists = st.mylist; *nl = 1;
      if (R > 1 || cm.comm) {
        MGX_RCCL(api.AllGather(st.mylist, bufs.glists.data(), (size_t)st.list_words(), ncclUint32, cm.comm, s));
        *lists = bufs.glists.data(); *nl = R;
      }
    };
    // ---- a whole traversal enqueued ahead, from what the last ones looked like (d2_run_bufs_t::plan): per level the push and
    // EITHER the lists (all-gather + merge) OR the bitmaps (exchange + merge) -- no host look in between.  Right whatever the
    // traversal does: bitmaps are always right; a list that overflows against the plan freezes the traversal at that level
    // (k_d2_lists_apply) and everything enqueued behind it returns at once; levels past the end find nothing.
    // AGREEMENT (round 6).  The plan decides which collective a rank issues per level, and every rank makes its own from its own
    // history: ranks whose histories differ -- an engine handle recreated on one of them, MGX_DIST_SPEC set in one environment,
    // a traversal that threw on some -- would enqueue different RCCL collectives: a hang, not an error.  So level 0 (the source:
    // always merged from the lists) runs with a host look on every rank, and its all-gathered list headers carry every rank's
    // plan: only when all are equal (k_d2_lists_apply -> host_flag[3], the same answer everywhere) is the rest enqueued ahead.
    bool over = false;
    bool agreed = false;
    {
      d2_push(st, 0, ctx);
      const u32* lists; int nl;
      gather_lists(&lists, &nl);
      long long o3[3];
      d2_apply_lists(st, 0, lists, nl, st.list_words(), ctx, o3);
      agreed = o3[2] != 0;
      level = 1;
      if (o3[1] == 0) {                                     // a source without edges: over
        MGX_CHECK_LAUNCH("partitioned BFS: kernel launch");
        d2_status(st, level, ctx, out6);
        bufs.learn(st.fs->host_ctrl, (int)out6[1], out6[3], st.list_cap);
        return;
      }
      if (o3[0]) d2_exchange_bitmaps(st, cm, bufs, 0, exchange, xwords, ctx);
    }
    if (have_plan && agreed) {
      bufs.spec_runs += 1;
      bufs.last_plan_levels = plan_levels; bufs.last_plan_sparse = plan_sparse;
      for (; level < plan_levels; ++level) {
        const bool sparse = level >= 64 || ((plan_sparse >> level) & 1ull);
        st.skip_small_reduce = sparse;
        d2_push(st, level, ctx, sparse);
        st.skip_small_reduce = false;
        if (sparse) {
          const u32* lists; int nl;
          gather_lists(&lists, &nl);
          (void)d2_enqueue_apply_lists(st, level, lists, nl, st.list_words(), ctx, true);
        } else {
          d2_exchange_bitmaps(st, cm, bufs, level, exchange, xwords, ctx);
        }
      }
      MGX_CHECK_LAUNCH("partitioned BFS: kernel launch");
      d2_status(st, level, ctx, out6);                    // (the one wait of a traversal that went as planned)
      const int frozen = st.fs->host_ctrl->d2_frozen_level;
      if (frozen >= 0) {
        // level `frozen` did not fit its lists: nothing of it was applied and nothing behind it ran.  Its new-bit map is intact:
        // the bitmaps now, then level by level with a look at each (the loop below)
        bufs.spec_frozen += 1;
        hipLaunchKernelGGL(k_d2_unfreeze, dim3(1), dim3(64), 0, s, st.args().ctrl);
        d2_exchange_bitmaps(st, cm, bufs, frozen, exchange, xwords, ctx);
        level = frozen + 1;
      } else if (out6[0]) {
        over = true;
      } else {
        bufs.spec_short += 1;                             // (a deeper traversal than any remembered: on, level by level)
      }
    }
    // ---- one level per round: id lists first; the bitmaps only when some rank's discoveries did not fit its list.  The host
    // looks at three words per level (d2_apply_lists) -- that is also how it learns that the traversal is over.  The first
    // traversal of an engine, MGX_DIST_SPEC=0, and what a plan left undone.
    while (!over) {
      d2_push(st, level, ctx);
      const u32* lists; int nl;
      gather_lists(&lists, &nl);
      long long o3[3];
      d2_apply_lists(st, level, lists, nl, st.list_words(), ctx, o3);
      if (o3[1] == 0) { ++level; break; }                   // nothing discovered on any rank: over
      if (o3[0]) d2_exchange_bitmaps(st, cm, bufs, level, exchange, xwords, ctx);
      ++level;
    }
    if (!over) {
      MGX_CHECK_LAUNCH("partitioned BFS: kernel launch");
      d2_status(st, level, ctx, out6);
    }
    bufs.learn(st.fs->host_ctrl, (int)out6[1], out6[3], st.list_cap);
    return;
  }
  int batch = bufs.levels_hint;
  for (;;) {
    for (int i = 0; i < batch; ++i, ++level) {
      d2_push(st, level, ctx);
      d2_exchange_bitmaps(st, cm, bufs, level, exchange, xwords, ctx);
    }
    MGX_CHECK_LAUNCH("partitioned BFS: kernel launch");
    d2_status(st, level, ctx, out6);
    if (out6[0]) break;
    batch = 2;
  }
  bufs.levels_hint = (int)(out6[1] > 0 ? out6[1] + 1 : 1);      // + the level that finds nothing
}

// ---- G rank engines of ONE partition on one device, driven in turn by one host thread ("group run") -------------------------
// What a rank's GPU does per traversal, measured without a second GPU: the engines share the stream, every collective is the G
// device copies that put the ranks' buffers side by side (ONE gathered buffer serves all engines: they are on the same device),
// and the level plan, the freeze and the level-by-level continuation are d2_run's.  Wall time / G = kernels + launch gaps of one
// rank per traversal, with no exchange time and no time of the other ranks in it -- tools/dist2_single.py's number without the
// interpreter between the launches.  Exchange 0 (all-gather) only; every engine needs an id list or none does.
struct d2_group_bufs_t {
  mem_t<u32> gathered, glists;
  long long xwords = 0;
};
inline void d2_group_run(d2_state_t** sts, d2_run_bufs_t** bufs, d2_group_bufs_t& gb, int G, int src, long long xwords,
                         standard_context_t& ctx, long long* out6 /* G x 6 */) {
  hipStream_t s = ctx.stream();
  d2_state_t& s0 = *sts[0];
  const bool lists = s0.mylist != nullptr;
  const long long lw = lists ? s0.list_words() : 0;
  if (gb.xwords != xwords || !gb.gathered.size()) {
    ctx.synchronize();
    gb.gathered = mem_t<u32>((size_t)G * (size_t)xwords + 4, ctx);
    gb.glists = mem_t<u32>((size_t)G * (size_t)(lw > 0 ? lw : 4) + 4, ctx);
    gb.xwords = xwords;
  }
  for (int r = 0; r < G; ++r) {
    int pl = 0; u64 ps = 0;
    const bool hp = lists && bufs[r]->plan(&pl, &ps);
    d2_reset(*sts[r], src, ctx, hp ? (u32)pl : 0u, hp ? ps : 0ull);
  }
  auto gather_lists = [&]() {
    for (int r = 0; r < G; ++r)
      MGX_HIP(hipMemcpyAsync(gb.glists.data() + (size_t)r * lw, sts[r]->mylist, (size_t)lw * sizeof(u32), hipMemcpyDeviceToDevice, s));
  };
  auto bitmaps = [&](int level) {
    for (int r = 0; r < G; ++r)
      MGX_HIP(hipMemcpyAsync(gb.gathered.data() + (size_t)r * xwords, sts[r]->newbits, (size_t)xwords * sizeof(u32), hipMemcpyDeviceToDevice, s));
    for (int r = 0; r < G; ++r) d2_merge(*sts[r], level, gb.gathered.data(), G, xwords, ctx);
  };
  int level = 0;
  bool over = false;
  if (lists) {
    int plan_levels = 0;
    u64 plan_sparse = 0;
    const bool have_plan = bufs[0]->plan(&plan_levels, &plan_sparse);
    // level 0 with a host look, as d2_run's agreement has it (the engines of a group share one history: the headers agree)
    bool agreed = true;
    {
      for (int r = 0; r < G; ++r) d2_push(*sts[r], 0, ctx);
      gather_lists();
      long long o3[3] = {0, 0, 0};
      for (int r = 0; r < G; ++r) { d2_apply_lists(*sts[r], 0, gb.glists.data(), G, lw, ctx, o3); agreed = agreed && o3[2] != 0; }
      level = 1;
      if (o3[1] == 0) {
        MGX_CHECK_LAUNCH("partitioned BFS (group): kernel launch");
        for (int r = 0; r < G; ++r) d2_status(*sts[r], level, ctx, out6 + 6 * r);
        for (int r = 0; r < G; ++r) bufs[r]->learn(sts[r]->fs->host_ctrl, (int)out6[6 * r + 1], out6[6 * r + 3], sts[r]->list_cap);
        return;
      }
      if (o3[0]) bitmaps(0);
    }
    if (have_plan && agreed) {
      for (int r = 0; r < G; ++r) { bufs[r]->spec_runs += 1; bufs[r]->last_plan_levels = plan_levels; bufs[r]->last_plan_sparse = plan_sparse; }
      for (; level < plan_levels; ++level) {
        const bool sparse = level >= 64 || ((plan_sparse >> level) & 1ull);
        for (int r = 0; r < G; ++r) { sts[r]->skip_small_reduce = sparse; d2_push(*sts[r], level, ctx, sparse); sts[r]->skip_small_reduce = false; }
        if (sparse) {
          gather_lists();
          for (int r = 0; r < G; ++r) (void)d2_enqueue_apply_lists(*sts[r], level, gb.glists.data(), G, lw, ctx, true);
        } else bitmaps(level);
      }
      MGX_CHECK_LAUNCH("partitioned BFS (group): kernel launch");
      for (int r = 0; r < G; ++r) d2_status(*sts[r], level, ctx, out6 + 6 * r);
      const int frozen = s0.fs->host_ctrl->d2_frozen_level;
      if (frozen >= 0) {
        for (int r = 0; r < G; ++r) { bufs[r]->spec_frozen += 1; hipLaunchKernelGGL(k_d2_unfreeze, dim3(1), dim3(64), 0, s, sts[r]->args().ctrl); }
        bitmaps(frozen);
        level = frozen + 1;
      } else if (out6[0]) over = true;
      else for (int r = 0; r < G; ++r) bufs[r]->spec_short += 1;
    }
    while (!over) {
      for (int r = 0; r < G; ++r) d2_push(*sts[r], level, ctx);
      gather_lists();
      long long o3[3] = {0, 0, 0};
      for (int r = 0; r < G; ++r) d2_apply_lists(*sts[r], level, gb.glists.data(), G, lw, ctx, o3);
      if (o3[1] == 0) { ++level; break; }
      if (o3[0]) bitmaps(level);
      ++level;
    }
    if (!over) {
      MGX_CHECK_LAUNCH("partitioned BFS (group): kernel launch");
      for (int r = 0; r < G; ++r) d2_status(*sts[r], level, ctx, out6 + 6 * r);
    }
    for (int r = 0; r < G; ++r) bufs[r]->learn(sts[r]->fs->host_ctrl, (int)out6[6 * r + 1], out6[6 * r + 3], sts[r]->list_cap);
    return;
  }
  int batch = bufs[0]->levels_hint;
  for (;;) {
    for (int i = 0; i < batch; ++i, ++level) {
      for (int r = 0; r < G; ++r) d2_push(*sts[r], level, ctx);
      bitmaps(level);
    }
    MGX_CHECK_LAUNCH("partitioned BFS (group): kernel launch");
    for (int r = 0; r < G; ++r) d2_status(*sts[r], level, ctx, out6 + 6 * r);
    if (out6[0]) break;
    batch = 2;
  }
  for (int r = 0; r < G; ++r) bufs[r]->levels_hint = (int)(out6[1] > 0 ? out6[1] + 1 : 1);
}

}  // namespace mgx
