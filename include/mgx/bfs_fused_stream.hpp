// mgx/bfs_fused_stream.hpp -- row-wise streaming push over the LONG-row queue of a level.
//
// The load-balanced search (k_bfs_push_level_wave, the reference's transform_lbs shape) resolves the row
// of EVERY edge rank: ~6 LDS probes and ~80 VALU instructions per edge, whatever the row length.  A skewed
// graph keeps most of its edges in long rows (RMAT-22: 82 % of the edges in rows of >= 64, 94 % in rows of
// >= 16), and for those nothing has to be searched: a wave reads 64 consecutive col_indices of ONE row per
// instruction.  So discoveries are split by degree when the next level's queues are built (k_bfs_build): rows of
// at least args.long_min edges go to the long-row queue, and this kernel streams them:
//
//   * the level's long rows are cut into equal contiguous slices of PADDED edge ranks (degrees rounded up to 64,
//     bfs_lq_* in bfs_fused.hpp: a unit is one sub-round below), one per wave (perfect balance
//     whatever the row lengths: a hub row spans many slices); one 64-ary search finds the slice's first row;
//   * the walk over (row, position) is wave-uniform and lives in SGPRs: a sub-round is up to 64 consecutive
//     edges of the current row, a round is EPT sub-rounds; the rows of the next round (at most EPT+1) are
//     prefetched as one coalesced load per array and picked with v_readlane;
//   * EPT col_indices loads per lane (EPT x 256 B per wave) are in flight while the previous round is tested:
//     64 KB per CU at 16 waves x 16 loads, what HBM needs (bytes in flight = bandwidth x latency);
//   * visited test: the hot prefix of the bitmap sits in LDS (160 KB = the 1.3 M highest-degree vertices under
//     the hub-first layout, ~97 % of RMAT-22's edge endpoints); a miss there claims the bit in LDS first (ds_or:
//     exact intra-workgroup dedup) and then stores mark[v] = 1.  Cold neighbours: either marked without a test
//     (COLDT = false: k_bfs_build tests the bitmap anyway; right when few endpoints are cold) or tested against
//     the L2-resident bitmap word, fetched with unconditional loads one round ahead (COLDT = true: big graphs);
//   * per edge: 1 coalesced load, ~10 VALU, 1 LDS read; no atomics, no staging, one barrier (after the
//     hot-bitmap copy).  Labels and the next level's queues are k_bfs_build's job.
#pragma once
#include "bfs_fused.hpp"

namespace mgx {

constexpr int BFS_STREAM_HOTW = 40896;     // words of the bitmap kept in LDS: 163584 B of the 160 KB

constexpr size_t bfs_stream_lds_bytes(int hotw) { return (size_t)hotw * 4 + 64; }

// DIAG: honour MGX_BFS_FLAGS (switch parts of the kernel off for measurements; results are then wrong by design).
// The kernel body as a device function: block `block` of `nblocks` (k_bfs_push / k_bfs_push_level in bfs_fused_run.hpp
// give it a part of a grid it shares with the other bodies).
//
// What bounds it (RMAT-22's big level, 98 M long edges: 151 us alone, 2.6 TB/s of col_indices): the latency of the
// memory pipeline under load.  8192 waves x 8 loads x 256 B = 16.8 MB in flight, returned in ~4 us -- the same ~4 TB/s
// tools/microbench3.hip gets from bare loads of this shape (profiles/r01/microbench3_inflight.jsonl), less the partial
// sub-rounds (88 % of the lanes carry an edge), the copy of the bitmap and the slice search in front.  Tried against
// that, each A/B'd on one box over the 16-source bench and each within +-1 % or slower: rounds that lie inside one
// row skipping the per-sub-round walk and lane masks; one ds_or with return instead of read-then-or; buffer loads
// (descriptor + scalar offset: no address arithmetic per sub-round); 16-byte loads for rounds inside one row, with
// and without 128-byte aligned starts (3 % slower); a ring that reloads a register right after its test so that
// seven loads stay in flight across rounds (2 % slower, although bare loads gain 20 % from it); 16 loads per lane
// and round instead of 8 (same).  What does move it: 16 extra VALU instructions per sub-round cost +28 us on that
// level (+13 %), 16 s_nop +15 us -- about half of any added issue time shows, so the ~47 instructions a sub-round
// costs now (~20 scalar for the walk, ~16 vector, the LDS probe, branches) are the lever, not the memory side.  The
// fast path for rounds inside one row did not help because that level has few of them: its long rows are the
// mid-degree ones (a few sub-rounds each); the hubs were expanded a level earlier.
template <int NT, int HOTW, int EPT, bool COLDT, bool DIAG = false, bool NTLOAD = false>
__device__ __forceinline__ void bfs_stream_body(const bfs_fused_args_t& a, int level, u32 block, u32 nblocks, int stat_level) {
  constexpr int NW = NT / WAVE;
  static_assert(EPT + 2 <= WAVE, "round shape");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32* const hot = (u32*)smem;
  int* const s_int = (int*)(hot + HOTW);                    // [0] marks stored
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);   // uniform: the walk stays in SGPRs
  const int lane = lane_id();

  bfs_ctrl_t* const c = a.ctrl;          // (`level` is the slot: ring and queue-buffer index, see bfs_resolve)
  const u64 cur = c->lcursor[level % 3];
  const u32 nf = (u32)(cur >> BFS_VSHIFT);
  const u32 E = (u32)(cur & BFS_EMASK);                     // in units of padded edges: a multiple of 64 (bfs_lq_*)
  if (nf == 0 || bfs_level_pulls(a, c, level)) return;

  const u32* __restrict__ q_row = a.lq_row[level & 1];
  const u32* __restrict__ q_off = a.lq_off[level & 1];
  const int* __restrict__ col = a.col_indices;
  const u32* __restrict__ vis = a.visited;
  unsigned char* __restrict__ mark = a.mark;

  // slice of this wave
  const u32 total_waves = nblocks * NW;
  u32 per = (E + total_waves - 1) / total_waves;
  per = (per + WAVE - 1) / WAVE * WAVE;
  const u64 rb = (u64)(block * NW + wave) * per;
  const bool has_work = rb < (u64)E;
  const u32 r_begin = has_work ? (u32)rb : E;
  const u32 r_end = (rb + per < (u64)E) ? (u32)(rb + per) : E;

  // hot prefix of the bitmap: 160 KB from L2 costs about a microsecond per CU
  const bool use_hot = E >= a.hot_min_edges;
  const u32 hot_n = use_hot ? (((u32)a.n < (u32)(HOTW * 32)) ? (u32)a.n : (u32)(HOTW * 32)) : 0u;
  if (use_hot) {
    bfs_copy_prefix<NT, HOTW>(hot, vis);
  }
  if (threadIdx.x == 0) s_int[0] = 0;
  __syncthreads();

  // marks of the vertices in [0, defer_n) wait for the end of the workgroup (bfs_hot_epilogue)
  const u32 defer_n = DIAG ? 0u : bfs_defer_limit(a, hot_n);
  const int diag = (DIAG && (MGX_LAB_GET(a, flags, 0) >> 8) == stat_level) ? (MGX_LAB_GET(a, flags, 0) & 255) : 0;   // MGX_BFS_FLAGS = level << 8 | bits
  int marks = 0;                 // per lane

  if (has_work) {
    // window registers: lane i holds queue entry (seg + i); entries past the queue read as (row 0, offset E)
    u32 w_row = 0, w_off = 0;
    auto prefetch = [&](u32 sg) {
      const u32 i = sg + (u32)lane;
      const bool ok = i < nf;
      const u32 ld_off = q_off[ok ? i : nf - 1];
      const u32 ld_row = q_row[ok ? i : nf - 1];
      w_off = ok ? ld_off : E;
      w_row = ok ? ld_row : 0u;
    };
    // first row of the slice
    u32 seg = (u32)(wave_upper_bound(q_off, (long long)nf, r_begin + 63u) - 1);     // (+63: the entries' low bits)
    prefetch(seg);
    u32 pos, end;                // col_indices range still to read of the current row (wave-uniform)
    {
      const u32 start = __builtin_amdgcn_readlane(w_row, 0);
      const u32 e0 = __builtin_amdgcn_readlane(w_off, 0);
      const u32 e1 = __builtin_amdgcn_readlane(w_off, 1);
      const u32 o0 = e0 & ~63u, o1 = e1 & ~63u;
      const u32 deg = bfs_lq_degree(e0, e1);
      const u32 lim = (o1 < r_end ? o1 : r_end) - o0;          // the slice may end inside the row (at a multiple of 64)
      pos = start + (r_begin - o0);
      end = start + (deg < lim ? deg : lim);
    }
    seg += 1;                    // next row to start
    prefetch(seg);
    bool fin = false;            // no further row starts inside the slice

    u32 baseL[EPT], nL[EPT];     // landing tile: where its loads read from (uniform)
    int idL[EPT];                // landing registers of the col_indices loads
    // one round of the walk: EPT sub-rounds of up to 64 consecutive edges of one row each
    u32 lastp = pos;             // some edge of this slice (has_work: there is one)
    auto walk = [&]() {
      int j = 0;
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        if (pos == end && !fin) {
          const u32 e0 = __builtin_amdgcn_readlane(w_off, j);
          const u32 o0 = e0 & ~63u;
          if (o0 >= r_end) {
            fin = true;
          } else {
            const u32 start = __builtin_amdgcn_readlane(w_row, j);
            const u32 e1 = __builtin_amdgcn_readlane(w_off, j + 1);
            const u32 o1 = e1 & ~63u;
            const u32 deg = bfs_lq_degree(e0, e1);
            const u32 lim = (o1 < r_end ? o1 : r_end) - o0;
            pos = start;
            end = start + (deg < lim ? deg : lim);
            ++j;
          }
        }
        const u32 left = end - pos;
        const u32 n = left < (u32)WAVE ? left : (u32)WAVE;
        baseL[k] = n ? pos : lastp;              // an empty sub-round re-reads (and re-tests) an edge of the slice
        lastp = n ? pos : lastp;
        nL[k] = n;
        pos += n;
      }
      seg += (u32)j;
    };
    auto issue = [&]() {
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const int* __restrict__ p = col + baseL[k];            // uniform base, 32-bit lane offset
        idL[k] = NTLOAD ? __builtin_nontemporal_load(p + (((u32)lane < nL[k]) ? (u32)lane : 0u))
                        : p[((u32)lane < nL[k]) ? (u32)lane : 0u];
      }
    };
    // test one round.  A hot miss claims the bit in the LDS copy (the bitmap + this workgroup's own marks): exact
    // intra-workgroup dedup.  (Probing all EPT words first and deciding afterwards was measured slower.)
    auto test_round = [&](const int (&id)[EPT], const u32 (&nn)[EPT], const u32* cold_word) {
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        const u32 d = (u32)id[k];
        const u32 bit = 1u << (d & 31);
        // lanes past the end of the sub-round hold the row's first neighbour again (issue() clamps them to lane 0):
        // testing it twice is harmless, and not masking them saves a compare and two mask operations per sub-round
        const bool act = !(diag & 16);                            // diag 16: stream only, no test
        const bool hotm = act && d < hot_n;
        // every lane probes (the others word 0): two nested branches less than "if active, if hot" per sub-round
        const u32 w = hot[hotm ? (d >> 5) : 0u];
        bool is_new = act && !hotm && (COLDT ? !(cold_word[k] & bit) : true);
        if (hotm && !(w & bit)) is_new = (diag & 2) || !(atomicOr(&hot[d >> 5], bit) & bit);
        if (is_new) {
          if (!(diag & 1) && d >= defer_n) mark[d] = 1;
          ++marks;
        }
      }
    };

    walk();
    issue();
    prefetch(seg);
    bool more = true;
    if constexpr (!COLDT) {
      // two stages: the loads of round t+1 are in flight while round t is tested
      int idT[EPT];
      u32 nT[EPT];
      while (more) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) { idT[k] = idL[k]; nT[k] = nL[k]; }
        more = !(fin && pos == end);
        walk();                  // needs the prefetched window: every load issued so far has landed
        issue();
        prefetch(seg);
        test_round(idT, nT, nullptr);
      }
    } else {
      // three stages: round t+1 loads its neighbours, round t the bitmap words of its cold ones, round t-1
      // is tested.  All loads unconditional (hot lanes read word 0: one broadcast request).
      int idT[EPT], idW[EPT];
      u32 nT[EPT], nW[EPT], wordT[EPT], wordL[EPT];
#pragma unroll
      for (int k = 0; k < EPT; ++k) { idW[k] = 0; nW[k] = 0; wordL[k] = 0xFFFFFFFFu; }
      bool moreW = true;         // a tile is waiting for its words
      bool haveT = false;        // the first pass has nothing to test yet (and no lane mask would hide the placeholders)
      while (more || moreW) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
          idT[k] = idW[k]; nT[k] = nW[k]; wordT[k] = wordL[k];
          idW[k] = idL[k]; nW[k] = nL[k];
        }
        moreW = more;
        more = more && !(fin && pos == end);
        walk();
        issue();
#pragma unroll
        for (int k = 0; k < EPT; ++k) wordL[k] = vis[((u32)idW[k] >= hot_n) ? ((u32)idW[k] >> 5) : 0u];
        prefetch(seg);
        if (haveT) test_round(idT, nT, wordT);
        haveT = true;
      }
    }
  }
  (void)bfs_hot_epilogue<NT>(a, hot, (defer_n + 31u) >> 5, level, s_int + 4, marks);
  bfs_body_finish(a, marks, level, stat_level, s_int);
}


}  // namespace mgx
