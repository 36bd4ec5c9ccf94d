// mgx/bfs_fused_dense.hpp -- push over the LONG rows of a level read from the graph's UNIT BLOCKS.
//
// bfs_fused_stream.hpp walks the level's long-row queue: per 64-edge sub-round ~20 scalar instructions of (row,
// position) bookkeeping, ~16 vector ones, a 4-byte-per-lane load whose address depends on the walk -- and the level
// that carries a skewed graph's work (RMAT-22: 116 M of 134 M edges) has nearly ALL long rows in its frontier (89 % of
// their edges), mid-degree ones, a few sub-rounds each, so the walk never amortises.  Here nothing is walked:
//
//   * the long rows live a second time as unit blocks (mgx_layout.hip): every row padded to a multiple of 64 entries
//     (padding entries are -1: they read as visited, see the sentinel words below), so a UNIT of 64 consecutive entries belongs to exactly one row,
//     owner[u] says which;
//   * a wave takes groups of 16 consecutive units (4 KB of entries), interleaved over all waves of the grid -- equal
//     shares whatever the frontier looks like (a level of a few thousand hubs activates one contiguous stretch of units);
//   * per batch of 4 groups: one coalesced load of 64 owners, one gather of their frontier bits (k_bfs_build leaves the
//     level's frontier as a bitmap), one ballot -> the 64 activity bits of the batch in two SGPRs.  Issued a batch
//     ahead: the two dependent loads never stall the stream;
//   * the entries are read 16 bytes per lane (1 KB per wave instruction, 4 units), four instructions per group, the next
//     active group's loads in flight while the last one is tested; a group without an active unit is skipped altogether,
//     lanes of inactive units in an active group read the four -1 behind the blocks instead (one cached line, no HBM
//     traffic, no divergent branch);
//   * visited test per entry: 3 VALU + one LDS read + 2 VALU on the fast path -- the LDS bitmap has a word of ones in
//     front (-1 entries read as visited) and a word of zeros behind (vertices outside the prefix read as unvisited), so
//     one clamp replaces the range checks; a miss claims the bit in LDS (exact intra-workgroup dedup) and stores
//     mark[v] = 1 as everywhere else.
//
// Chosen per slot and by every workgroup alike (bfs_long_is_dense): the frontier bitmap must describe this slot's
// frontier (ctrl->fb_slot) and the frontier must hold at least 1 / dense_div of the units; otherwise the queue walk of
// bfs_fused_stream.hpp runs.  Labels are the same either way: both mark exactly the unvisited neighbours of the
// frontier's long rows.
#pragma once
#include <type_traits>
#include "bfs_fused.hpp"

namespace mgx {

typedef unsigned int bfs_u32x4 __attribute__((ext_vector_type(4)));   // (a builtin vector: __builtin_nontemporal_load wants one)
typedef unsigned int bfs_u32x3 __attribute__((ext_vector_type(3)));   // four 24-bit entries (global_load_dwordx3: 4-byte alignment is all the hardware asks)

// four 24-bit entries out of three words, sign-extended (0xFFFFFF -> -1: padding and the lanes of inactive units): 6 vector
// instructions per four entries -- against a quarter of the level's HBM bytes
__device__ __forceinline__ bfs_u32x4 bfs_unpack24(bfs_u32x3 w) {
  bfs_u32x4 d;
  d.x = (u32)__builtin_amdgcn_sbfe((int)w.x, 0, 24);
  d.y = (u32)__builtin_amdgcn_sbfe((int)__builtin_amdgcn_alignbit(w.y, w.x, 24), 0, 24);
  d.z = (u32)__builtin_amdgcn_sbfe((int)__builtin_amdgcn_alignbit(w.z, w.y, 16), 0, 24);
  d.w = (u32)((int)w.z >> 8);
  return d;
}

constexpr int BFS_DENSE_GROUP = 16;                 // units per group (the unit of interleaving)
constexpr size_t bfs_dense_lds_bytes(int hotw) { return (size_t)hotw * 4 + 128; }

// the slot's long rows are read from the unit blocks (grid-uniform: every workgroup sees the same stable inputs)
__device__ __forceinline__ bool bfs_long_is_dense(const bfs_fused_args_t& a, const bfs_ctrl_t* c, int slot, u64 lcur) {
  if (!a.ub_owner || a.dense_div == 0u || c->fb_slot != slot) return false;      // (ub_col may be NULL: a layout that carries the 24-bit copy only)
  const u64 units = (lcur & BFS_EMASK) >> 6;        // the long-row queue's offsets count padded edges: 64 per unit
  return units * (u64)a.dense_div >= (u64)a.ub_units;
}

// The LDS prefix of the visited bitmap as the unit-block and vertex-by-vertex bodies use it: a word of ones in front
// (hot[-1]: the -1 entries of inactive lanes read as visited), a word of zeros behind (hot[HOTW]: vertices outside the
// prefix read as unvisited), so that one clamp replaces the range checks of the probe.  Returns hot; all threads.
template <int NT, int HOTW>
__device__ __forceinline__ u32* bfs_hot_setup(const bfs_fused_args_t& a, char* smem, int** s_int, u32 behind = 0u) {
  u32* const hot = (u32*)smem + 4;
  *s_int = (int*)(hot + HOTW + 4);
  bfs_copy_prefix<NT, HOTW>(hot, a.visited);
  // (behind: what a vertex outside the prefix reads as -- 0 "unvisited": marked untested; all ones when the slot's cold-edge
  //  pass takes care of those entries, bfs_fused_cold.hpp)
  if (threadIdx.x == 0) { hot[-1] = 0xFFFFFFFFu; hot[HOTW] = behind; (*s_int)[0] = 0; }
  __syncthreads();
  return hot;
}

// the blocks a pass reads: entries at 32 or at 24 bits, owners, units (padded to a multiple of 16)
struct bfs_units_view_t {
  const int* col;
  const u32* col24;
  const int* owner;
  u32 units_pad;
};

// the unit-block pass of one workgroup (block `block` of `nblocks`) over an LDS prefix that is already set up
// P24: the entries come from a 24-bit copy of the unit blocks (ub.col24: the full blocks of a graph of at most 2^23 vertices, or the
// blocks without the cold-edge lists' entries, whose ids lie inside the LDS prefix): 12 bytes per lane and load instead of 16,
// unpacked when they are tested

template <int NT, int HOTW, int GPS, bool P24 = false>
__device__ __forceinline__ void bfs_dense_work(const bfs_fused_args_t& a, const bfs_units_view_t ub, u32* const hot, u32 hot_n, u32 defer_n, u32 block,
                                               u32 nblocks, int& marks) {
  constexpr int NW = NT / WAVE;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
  const int lane = lane_id();
  unsigned char* __restrict__ mark = a.mark;
  const int* __restrict__ ucol = ub.col;
  const int* __restrict__ owner = ub.owner;
  const u32* __restrict__ fbits = a.frontier_bits;
  const int diag = MGX_LAB_GET(a, dense_diag, 0);     // MGX_BFS_DENSE_DIAG (measurements; results are wrong by design): 1 no mark stores, 2 no test
  const u32 G = ub.units_pad / BFS_DENSE_GROUP;                // groups of 16 units
  const u32 W = nblocks * NW;                                  // waves of the grid
  const u32 w = block * NW + (u32)wave;
  const u32 dummy = ub.units_pad << 6;                         // entry index of the four -1
  const u32 lane_unit = (u32)lane & 15u;                       // my unit inside its group, for the owner load ...
  const u32 lane_grp = (u32)lane >> 4;                         // ... and which of the batch's 4 groups
  const u32 lane_q = (u32)lane >> 4;                           // col loads: lane l reads entries 4l..4l+3 of a 256-entry chunk = unit l >> 4 of the chunk

  // batch b of this wave: groups w + (4 b + k) W, k = 0..3
  const u32 nbatch = (w < G) ? ((G - w + W - 1) / W + 3) / 4 : 0u;
  if (nbatch != 0u) {
    auto load_owner = [&](u32 b) -> u32 {
      const u32 g = w + (4u * b + lane_grp) * W;
      const u32 u = (g < G ? g : G - 1u) * BFS_DENSE_GROUP + lane_unit;
      const u32 o = (u32)owner[u];
      return g < G ? o : (u32)a.n;                             // vertex n: never in a frontier (its bit exists and stays 0)
    };
    auto gather_bits = [&](u32 own) -> u32 { return fbits[own >> 5]; };
    // entry index of the first entry of group k of batch b
    auto group_base = [&](u32 b, u32 k) -> u32 {
      const u32 g = w + (4u * b + k) * W;
      return (g < G ? g : 0u) * (BFS_DENSE_GROUP * 64u);
    };

    constexpr int NL = 4 * GPS;               // loads per step
    typedef typename std::conditional<P24, bfs_u32x3, bfs_u32x4>::type raw_t;     // what a lane's load returns
    raw_t dL[NL], dT[NL];
    const u32* __restrict__ ucol24 = ub.col24;
    // issue the 16-byte loads of groups k .. k + GPS - 1 of a batch whose activity bits are `act`
    auto issue = [&](u32 b, u32 k, u64 act) {
#pragma unroll
      for (int g = 0; g < GPS; ++g) {
        const u32 base = group_base(b, k + (u32)g);
        const u32 bits16 = (u32)(act >> (16u * (k + (u32)g))) & 0xFFFFu;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool on = (bits16 >> (4u * j + lane_q)) & 1u;
          const u32 e = on ? base + (u32)j * 256u + (u32)lane * 4u : dummy;
          if constexpr (P24) dL[4 * g + j] = __builtin_nontemporal_load((const bfs_u32x3*)(ucol24 + (size_t)(e >> 2) * 3u));   // (e is a multiple of 4)
          else dL[4 * g + j] = __builtin_nontemporal_load((const bfs_u32x4*)(ucol + e));
        }
      }
    };
    auto probe = [&](u32 d) -> u32 {
      int idx = (int)d >> 5;
      idx = idx < -1 ? -1 : idx;
      idx = idx > HOTW ? HOTW : idx;
      return hot[idx];
    };
    // A miss (bit not set in the workgroup's LDS copy): set it there -- a NON-returning LDS OR: nobody waits for the old
    // word -- and, outside the deferred range, store the mark.  The claim used to be a returning atomicOr whose old value
    // decided whether the mark was stored (exact dedup inside the workgroup): per entry a dependent LDS round trip, five
    // more vector instructions and two more divergent regions, executed by nearly every wave (3 % of a big level's entries
    // miss, i.e. 86 % of its 64-lane instructions hold one) -- the kernel was issue-bound on them (PMC, round 4: 18 vector
    // instructions per entry, VALU busy 45 % of a launch that streams at 60 % of the load rate).  Marks are idempotent byte
    // stores, so the only thing lost is the dedup of the few probes that are in flight between another lane's probe and
    // its OR: `marks` now counts misses (still an upper bound of the discoveries, which is all its readers ask of it).
    auto decide = [&](u32 d, u32 wd) {
      if (!((wd >> (d & 31u)) & 1u)) {
        if (d < hot_n) (void)__hip_atomic_fetch_or(&hot[d >> 5], 1u << (d & 31u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!(diag & 1) && d >= defer_n && !((diag & 4) && d >= hot_n)) mark[d] = 1;       // (diag 4: no COLD marks)
        ++marks;
      }
    };
    // four LDS probes in flight, then the four decisions (a probe that waits for its own result before the next one
    // is issued leaves the LDS pipe idle: 16 dependent round trips per stage)
    auto test = [&]() {
      if (diag & 2) {                  // measurement only: consume the loads, test nothing
#pragma unroll
        for (int j = 0; j < NL; ++j) {
          if constexpr (P24) marks += (int)((dT[j].x ^ dT[j].y ^ dT[j].z) == 0x12345678u);
          else marks += (int)((dT[j].x ^ dT[j].y ^ dT[j].z ^ dT[j].w) == 0x12345678u);
        }
        return;
      }
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        bfs_u32x4 d;
        if constexpr (P24) d = bfs_unpack24(dT[j]); else d = dT[j];
        const u32 w0 = probe(d.x), w1 = probe(d.y), w2 = probe(d.z), w3 = probe(d.w);
        decide(d.x, w0); decide(d.y, w1); decide(d.z, w2); decide(d.w, w3);
      }
    };

    // A group of 16 units whose activity bits are all 0 costs nothing: no load, no test.  The loads of the last active
    // group stay in flight until the next active one is found (or the wave runs out of groups): wait for the old,
    // issue the new, test the old.
    bool pend = false;
    auto step = [&](u32 b, u32 k, u64 act) {
      const u64 mask = GPS == 2 ? 0xFFFFFFFFull : 0xFFFFull;
      if (((act >> (16u * k)) & mask) == 0ull) return;     // wave-uniform
      if (pend) {
#pragma unroll
        for (int j = 0; j < NL; ++j) dT[j] = dL[j];
      }
      issue(b, k, act);
      if (pend) test();
      pend = true;
    };
    u32 own_next = load_owner(0);
    u64 act;
    {
      const u32 fw = gather_bits(own_next);
      act = __ballot((fw >> (own_next & 31u)) & 1u);
    }
    own_next = load_owner(1);
    for (u32 b = 0; b < nbatch; ++b) {
      // activity of batch b + 1: its owners were loaded a batch ago
      const u32 own_cur = own_next;
      const u32 fw = gather_bits(own_cur);
      own_next = load_owner(b + 2);
#pragma unroll
      for (u32 k = 0; k < 4; k += GPS) step(b, k, act);
      act = __ballot((fw >> (own_cur & 31u)) & 1u);
    }
    if (pend) {
#pragma unroll
      for (int j = 0; j < NL; ++j) dT[j] = dL[j];
      test();
    }
  }
}

// GPS: groups per step (1: four 16-byte loads in flight per lane while the previous four are tested; 2: eight -- for
// launches with half the waves per CU)
template <int NT, int HOTW, int GPS = 1>
__device__ __forceinline__ void bfs_dense_body(const bfs_fused_args_t& a, int slot, u32 block, u32 nblocks, int stat_level,
                                               bool cold = false) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_int;
  u32* const hot = bfs_hot_setup<NT, HOTW>(a, smem, &s_int, cold ? 0xFFFFFFFFu : 0u);    // cold: those entries are somebody else's
  const int lane = lane_id();
  bfs_ctrl_t* const c = a.ctrl;
  const u32 hot_n = ((u32)a.n < (u32)(HOTW * 32)) ? (u32)a.n : (u32)(HOTW * 32);
  const u32 defer_n = bfs_defer_limit(a, hot_n);      // marks of the vertices in [0, defer_n) wait for the end of the workgroup
  int marks = 0;
  // (grid-uniform) the blocks without the cold-edge lists' entries are for the slots that run the cold-edge pass
  bfs_units_view_t ub{a.ub_col, a.ub_col24, a.ub_owner, a.ub_units_pad};
  if (a.ub_hot_only && !cold) { ub.col24 = a.ubf_col24; ub.owner = a.ubf_owner; ub.units_pad = a.ubf_units_pad; }
  if (ub.col24) bfs_dense_work<NT, HOTW, GPS, true>(a, ub, hot, hot_n, defer_n, block, nblocks, marks);
  else bfs_dense_work<NT, HOTW, GPS, false>(a, ub, hot, hot_n, defer_n, block, nblocks, marks);
  (void)bfs_hot_epilogue<NT>(a, hot, (defer_n + 31u) >> 5, slot, s_int + 4, marks);
  bfs_body_finish(a, marks, slot, stat_level, s_int);
}

}  // namespace mgx
