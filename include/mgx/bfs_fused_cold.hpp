// mgx/bfs_fused_cold.hpp -- the COLD entries of the long rows: pairs by slice, a slice of the bitmap in LDS, no marks.
//
// The unit-block body (bfs_fused_dense.hpp) tests an entry against the first 652 288 vertices of the visited bitmap, in
// LDS.  An entry that points behind them ("cold": 7 % of the long rows' entries on RMAT-22) cannot be tested there; it was
// marked untested, one byte, wherever it pointed.  Measured (MGX_BFS_DENSE_DIAG=4, tools/microbench4.hip): those stores
// cost the big level 37-44 of its 150-190 us -- a scattered store into a mark array that does not stay in L2 under the
// stream costs the fabric as much as a whole cache line (85 G/s alone, and it halves a 16-byte-per-lane stream at one
// store per 16 loaded entries), and the same vertex is marked five times from different workgroups.
//
// So the cold entries are pulled out of the rows when the layout is built (mgx_layout.hip: mgx_cold_build_device): pairs
// (owner, dst), grouped by the SLICE of the id range dst lies in -- a slice is as many vertices as the LDS prefix holds --
// and ordered by owner inside a slice.  On a level that reads the unit blocks, a few hundred workgroups of the same push
// launch take the pairs instead (the unit-block body then skips cold entries: its sentinel word behind the prefix reads
// "visited"): a workgroup copies ITS slice of the bitmap into LDS, streams its share of the slice's pairs -- 8 bytes per
// pair, coalesced -- asks the frontier bitmap for the owner (neighbouring lanes, neighbouring or equal owners), tests and
// claims dst in LDS.  Every cold endpoint is now TESTED, duplicates die in LDS, and what is left at the end is a bitmap:
// the difference to the slice it started from goes to the workgroup's own flush buffer (80 KB, coalesced), and
// k_bfs_build2 ORs the buffers of a slice into its sweep as it does with the deferred marks of the prefix.  No scattered store is left on the long rows' side
// of such a level.
#pragma once
#include "bfs_fused.hpp"

namespace mgx {

constexpr size_t bfs_cold_lds_bytes() { return (size_t)BFS_COLD_WORDS * 4 + 128; }

// cold workgroup `cw` (0 .. cold_wgs[cold_slices] - 1) of slot `slot`; all threads
template <int NT>
__device__ __forceinline__ void bfs_cold_body(const bfs_fused_args_t& a, int slot, u32 cw, int stat_level, bool do_long, bool do_short) {
  constexpr int HOTW = BFS_COLD_WORDS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  u32* const hot = (u32*)smem + 4;
  int* const s_int = (int*)(hot + HOTW + 4);
  int sl = 0;
  while (sl + 1 < a.cold_slices && cw >= a.cold_wgs[sl + 1]) ++sl;             // (uniform)
  if (cw >= a.cold_wgs[a.cold_slices]) return;            // (cannot happen: the slices share all cold workgroups)
  const u32 part = cw - a.cold_wgs[sl], parts = a.cold_wgs[sl + 1] - a.cold_wgs[sl];
  const u32 lo = a.cold_lo[sl];                      // first vertex of the slice: a multiple of 1024
  const u32 w0 = lo >> 5;
  const u32 nwords = ((u32)a.n + 31u) >> 5;
  const u32 have = w0 < nwords ? (nwords - w0 < (u32)HOTW ? nwords - w0 : (u32)HOTW) : 0u;   // words of the slice that exist
  {
    // all loads first (bfs_copy_prefix's reason), 16 bytes each (w0 is a multiple of 32 words); words behind the bitmap's end
    // read as "visited"
    constexpr int Q = HOTW / 4, IT = (Q + NT - 1) / NT;
    const uint4* const s4 = (const uint4*)(a.visited + w0);
    const u32 q_have = (have + 3u) / 4u;                     // 16-byte pieces that hold existing words (the bitmap is padded to whole pieces)
    uint4 v[IT];
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      const u32 i = (u32)k * NT + threadIdx.x;
      v[k] = s4[i < q_have ? i : 0u];
    }
#pragma unroll
    for (int k = 0; k < IT; ++k) {
      const u32 i = (u32)k * NT + threadIdx.x;
      if (i < (u32)Q) {
        uint4 x = v[k];
        if (i * 4u + 0u >= have) x.x = 0xFFFFFFFFu;
        if (i * 4u + 1u >= have) x.y = 0xFFFFFFFFu;
        if (i * 4u + 2u >= have) x.z = 0xFFFFFFFFu;
        if (i * 4u + 3u >= have) x.w = 0xFFFFFFFFu;
        ((uint4*)hot)[i] = x;
      }
    }
  }
  if (threadIdx.x == 0) { hot[-1] = 0xFFFFFFFFu; hot[HOTW] = 0xFFFFFFFFu; s_int[0] = 0; s_int[1] = 0; }
  __syncthreads();

  const u32* __restrict__ fbits = a.frontier_bits;
  int marks = 0;
  constexpr int K = 4;                               // pairs per thread and round: 8 loads in flight, then 4 gathers
  // the long rows' list of the slice, then the short rows' (whichever the slot's bodies leave to this pass)
  for (int which = 0; which < 2; ++which) {
    if (which == 0 ? !do_long : !do_short) continue;
    // (the short rows' lists: graphs of more than 2^23 vertices, or MGX_BFS_COLD_LISTS=2 when the layout is built)
    const int* __restrict__ owner = which == 0 ? a.cold_owner : a.colds_owner;
    const int* __restrict__ dst = which == 0 ? a.cold_dst : a.colds_dst;
    const u32 p0 = which == 0 ? a.cold_off[sl] : a.colds_off[sl], p1 = which == 0 ? a.cold_off[sl + 1] : a.colds_off[sl + 1];
    const u32 chunk = (((p1 - p0) + parts - 1u) / parts + 255u) & ~255u;
    const u32 b = p0 + part * chunk < p1 ? p0 + part * chunk : p1;
    const u32 e = b + chunk < p1 ? b + chunk : p1;
    // four bytes per pair where the slice has them (args.cold_pk): the pair's word and the owner of its 64-chunk -- one address
    // per wave (b is a multiple of 256 pairs behind p0: a wave's 64 pairs are one chunk)
    const bool packed = which == 0 && a.cold_pk != nullptr && ((a.cold_pk_mask >> sl) & 1ull);
    const u32* __restrict__ pk = a.cold_pk;
    const u32* __restrict__ cbase = a.cold_cbase + a.cold_cb[sl];
    for (u32 r0 = b; r0 < e; r0 += (u32)NT * K) {
      const u32 base = r0 + threadIdx.x;
      u32 ow[K], dd[K];
      if (packed) {
        u32 pw[K], cb[K];
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const u32 i = base + (u32)k * NT;
          const u32 j = i < e ? i : b;               // (a pair of this workgroup's range: readable, ignored)
          pw[k] = pk[j];
          cb[k] = cbase[(j - p0) >> 6];
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
          const bool in = base + (u32)k * NT < e;
          ow[k] = in ? cb[k] + (pw[k] >> 20) * a.cold_ranks : 0xFFFFFFFFu;
          dd[k] = in ? lo + (pw[k] & 0xFFFFFu) : 0xFFFFFFFFu;
        }
      } else {
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const u32 i = base + (u32)k * NT;
        const bool in = i < e;
        const u32 j = in ? i : p1;                   // (p1 .. p1 + 255: the next slice's pairs or the padding: readable, ignored)
        ow[k] = (u32)owner[j];
        dd[k] = in ? (u32)dst[j] : 0xFFFFFFFFu;
        if (!in) ow[k] = 0xFFFFFFFFu;
      }
      }
      u32 fw[K];
#pragma unroll
      for (int k = 0; k < K; ++k) fw[k] = fbits[ow[k] != 0xFFFFFFFFu ? ow[k] >> 5 : 0u];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const bool act = ow[k] != 0xFFFFFFFFu && ((fw[k] >> (ow[k] & 31u)) & 1u);
        if (act) {
          const u32 r = dd[k] - lo;                  // < HOTW * 32 by construction
          const u32 bit = 1u << (r & 31u);
          if (r < (u32)HOTW * 32u && !(hot[r >> 5] & bit)) {
            if (!(atomicOr(&hot[r >> 5], bit) & bit)) ++marks;
          }
        }
      }
    }
  }

  // ---- what this workgroup discovered: a bitmap (or a handful of marks) -----------------------------------------------
  constexpr int PERT = (HOTW + NT - 1) / NT;
  __syncthreads();
  u32 diff[PERT];
#pragma unroll
  for (int q = 0; q < PERT; ++q) {
    const u32 i = (u32)q * NT + threadIdx.x;
    diff[q] = i < have ? (hot[i] & ~a.visited[w0 + i]) : 0u;
  }
  // always the bitmap (zeros included): the queue build ORs every buffer of a slice without asking which ones matter
  {
    u32* const out = a.cold_flush + (size_t)cw * HOTW;
#pragma unroll
    for (int q = 0; q < PERT; ++q) {
      const u32 i = (u32)q * NT + threadIdx.x;
      if (i < (u32)HOTW) out[i] = diff[q];
    }
  }
  bfs_body_finish(a, marks, slot, stat_level, s_int);
}

}  // namespace mgx
