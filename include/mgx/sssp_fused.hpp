// mgx/sssp_fused.hpp -- device-resident SSSP (frontier Bellman-Ford, the reference's sssp_enactor_t loop,
// sssp_enactor.hxx:40-72) without a host round trip per iteration and without the per-edge output frontier.
//
// The operator path (advance writing one slot per EDGE, then a compaction that reads them back; two size
// read-backs per iteration) spends 7 ms of an 8.6 ms RMAT-22 run in the advance kernel.  Here an iteration is
//   (bookkeeping: one thread of the relax launch, sssp_open)
//   k_sssp_relax   the frontier's edges, load-balanced per edge rank (wave-private, same scheme as
//                  bfs_fused_wave.hpp); per edge: neighbour and weight (coalesced),
//                  candidate = distance the row entered the frontier with + weight, one gather of the neighbour's
//                  distance; only an improving candidate issues the atomicMin (on the integer view of the
//                  non-negative float: order preserving, intrinsics.hxx:12-22 does a CAS loop) and stores mark[v] = 1
//   k_sssp_build   sweep over the marks: marked vertices form the next frontier -- (row start, exclusive degree
//                  scan, current distance) appended in batches with ONE packed 64-bit cursor atomic (bfs_fused.hpp);
//                  marks are cleared on the way.  A vertex improved several times in an iteration enters once: the
//                  reference's filter (stamp dedup, sssp_functor.hxx cond_filter) comes for free.
// Near / far buckets (delta > 0; the delta-stepping BASELINE config 3 names, in the near-far form): the queue build
// only takes improved vertices whose distance lies below the current threshold T; the others KEEP their mark and wait.
// When an iteration finds the near queue empty and marks left, T moves to the bucket of the smallest waiting distance
// and the build sweeps again.  A vertex far from the source is then expanded once its distance has (nearly) settled
// instead of once per improvement: plain Bellman-Ford relaxes 1.5-2.2 x the edges a settled-order run needs on RMAT-22.
// The fixpoint of min-plus relaxation is unique whatever the order of the relaxations (float addition is
// monotone), so the distances are bit-identical to the reference algorithm's.  Predecessors are not maintained
// (the reference's are racy); the operator path keeps them.
// Tried and dropped: a second queue for rows of >= 64 edges with a row-wise streaming relax kernel (the BFS design).
// The relaxation rate barely moved (86 vs 80 G/s: the kernel is bound by the distance gather and the atomics, not
// by the row search) and processing the hubs in a kernel of their own cost 1.4 x the relaxations: 3.36 vs 2.81 ms.
// Also dropped (round 3): relaxing WITHOUT the look at dist[dst] while few vertices have a distance (the first heavy iteration
// of a run from the hubs: nearly every candidate is the first for its vertex, so the gather looked like a wasted random
// access) -- atomicMin at once, the mark by its return value.  RMAT-22: 3.3-4.4 ms per source against 1.86; the iteration
// it was meant for went from 0.39 to 1.94 ms (15 M edges) and from 0.89 to 3.3 ms (70 M).  A dozen edges lead to every new
// vertex in that iteration: the look lets all but the first few skip the atomic (the line is in the L2 by then), and
// atomics are what this part is slowest at -- 8-21 G/s when every edge issues one, an order below its gathers.
// And the heaviest iterations BOTTOM-UP (on a graph whose weights the library had verified to be symmetric: every vertex folds
// min(dist[u] + w) over its own row -- the (min, +) twin of mgx/nreduce.hpp: unit blocks, degree classes, the hubs' exact
// distances in LDS; no atomics for short rows, one pre-checked atomicMin per unit of a long row; distances bit-equal in the
// tests): 2.27-2.30 ms per RMAT-22 source against 1.88.  A sweep costs 0.71-0.75 ms (12 bytes per edge and a gather, every
// edge every time) where the push sweep takes 0.75 once and 0.35-0.45 after, and it converges more slowly -- a push iteration
// passes an improvement on within the iteration (dist[u] is read when the row is staged, the atomicMin lands at once), a pull
// iteration works from the distances the workgroup copied at its start: 13 iterations and 473 M relaxation-equivalents against
// 10 and 297 M for the same source.
// And a HYBRID of the two for the hubs' iteration only (the early iterations with >= 2 M frontier edges, symmetric weights): the
// pushes -- queue walk and sweep -- skip destinations that are short rows (< 64 entries: four in five of that iteration's atomics
// go there) and one pass over the short rows lets each fetch min(dist[u] + w) over its frontier neighbours itself (frontier bits
// of the first 655 K vertices in LDS, a plain store per improved vertex).  Distances bit-equal; 1.90-1.93 ms per source against
// 1.85-1.88: the iteration of a source whose hubs hold 70 M edges went from 0.85 to 0.73 ms, the one with 15 M edges from 0.38 to
// 0.44 -- the pass over all 18 M short-row entries costs ~0.15 ms whatever the frontier (two dependent gathers per entry), more
// than the atomics it replaces.
// And (round 5) the heavy iterations over the long rows regrouped BY SLICE OF THEIR DESTINATIONS -- the layout the neighbour-reduce
// got in the same round (mgx/nreduce.hpp: 16-byte mini-units of eight 16-bit offsets, + eight weights and the row per mini-unit
// here): a workgroup keeps one slice's 40 000 distances in LDS as minima it maintains itself (a read per candidate, ds_min only
// when lower), and writes what became smaller back with ONE atomicMin per vertex when it leaves the slice; the tail behind the 16
// hot slices and the short rows stayed with the sweep kernel.  Distances bit-equal in every test (R-MAT 17 with 1 / 2 / 4 hot
// slices, integer and real weights).  RMAT-22: 2.03-2.23 ms per source against 1.77 whatever share of the edges an iteration had
// to hold to take it (profiles/r05/sssp_sliced_ab.log): the iteration with nearly all edges went from 0.70-0.81 to 0.57-0.63 ms,
// but a minimum that sits in a workgroup's LDS is invisible to the rows other workgroups stage in the same iteration, so the
// improvements arrive an iteration later: 12 iterations instead of 10, the fourth with 84 M instead of 72 M edges, a fifth with 22 M
// instead of 0.4 M.  The push loop lives on improvements landing AT ONCE (dist[u] is read when a row is staged).  The code is in
// the history of this file (commits da33246, 3c75a75 of round 5).
#pragma once
#include <hip/hip_fp16.h>
#include <type_traits>
#include <vector>
#include "bfs_fused.hpp"

namespace mgx {

struct sssp_args_t {
  const u32* row_offsets;
  const int* col_indices;
  const float* weights;
  u32* dist;             // float bits, n entries
  unsigned char* mark;   // n bytes (+ padding)
  u32* q_row[2];         // frontier queue: CSR row start
  u32* q_off[2];         //                 exclusive degree scan
  u32* q_du[2];          //                 the vertex id: its distance is read when the row is staged, so improvements
                         //                 made earlier in the same iteration are propagated (the operator path reads
                         //                 labels[src] per edge, sssp_functor.hxx cond_advance: same effect, 1.6-2 x fewer
                         //                 relaxations than with the distance frozen at queue time)
  bfs_ctrl_t* ctrl;      // cursor[3] (packed, rotating), sums, done, levels = iterations
  int n;
  u32 hot_min_edges;     // iterations with at least this many edges keep distance bounds of the hubs in LDS
  unsigned long long m_edges;   // edges of the graph (the test for a heavy iteration: frontier edges x dense_div >= m_edges)
  u32* frontier_bits;           // the frontier as a bitmap (k_sssp_build2 writes it, the sweep reads it; NULL: no sweep)
  // heavy iterations over the layout's unit blocks and degree classes (sssp_dense_*; NULL: never): the long rows' entries
  // and weights in 64-entry units of one row each, real entries per unit, the row of every unit; the short rows' classes
  const int* ub_col;
  const float* ub_w;
  const unsigned char* ub_cnt;
  const int* ub_owner;
  // the same entries as 24-bit ids and the same weights as IEEE halves (round 4; both or neither: graphs of at most 2^23
  // vertices whose weights are ALL exactly representable in 16 bits -- small integers, for one): 5 instead of 8 bytes per
  // entry of the sweep's stream, the arithmetic unchanged (a half converts to the float it was made from)
  const u32* ub_col24;
  const unsigned short* ub_w16;
  u32 ub_units_pad;
  u32 vs_v[4];
  u32 dense_div;         // an iteration whose frontier holds >= m / dense_div edges takes the sweep (0: never)
  float delta;           // near / far bucket width (delta-stepping; BASELINE config 3 names it): 0 = plain frontier
                         // Bellman-Ford, every improved vertex is expanded in the next iteration
};

// layout (optional): hub-first relabelled CSR with its weights and the two id maps; distances are reported in
// original ids either way
struct sssp_layout_t {
  const int* row_offsets = nullptr;
  const int* col_indices = nullptr;
  const float* weights = nullptr;
  const int* new_of_old = nullptr;
  const int* old_of_new = nullptr;
  long long m_edges = 0;
  // unit blocks of this CSR with their weights, degree classes (optional: a layout the library built and sorted itself)
  const int* ub_col = nullptr;
  const float* ub_w = nullptr;
  const unsigned char* ub_cnt = nullptr;
  const int* ub_owner = nullptr;
  const unsigned* ub_col24 = nullptr;            // 24-bit entries / half weights (see sssp_args_t); NULL: not available
  const unsigned short* ub_w16 = nullptr;
  unsigned ub_units_pad = 0;
  unsigned vs_v[4] = {0, 0, 0, 0};
};

constexpr u32 SSSP_INF_BITS = 0x7f7fffffu;      // FLT_MAX: what the reference stores for "not reached" (sssp_problem.hxx:45)

__global__ __launch_bounds__(BLOCK) void k_sssp_init(sssp_args_t a, int src, const int* __restrict__ new_of_old) {
  if (new_of_old) src = new_of_old[src];               // the loop runs in layout space
  const long long tid = (long long)blockIdx.x * BLOCK + threadIdx.x;
  const long long nth = (long long)gridDim.x * BLOCK;
  const long long n = a.n;
  for (long long i = tid; i < (n + 3) / 4; i += nth) {
    const long long j = i * 4;
    if (j + 4 <= n) {
      *(uint4*)(a.dist + j) = make_uint4(src == j ? 0u : SSSP_INF_BITS, src == j + 1 ? 0u : SSSP_INF_BITS,
                                         src == j + 2 ? 0u : SSSP_INF_BITS, src == j + 3 ? 0u : SSSP_INF_BITS);
      *(u32*)(a.mark + j) = 0u;
    } else {
      for (long long k = j; k < n; ++k) { a.dist[k] = (k == src) ? 0u : SSSP_INF_BITS; a.mark[k] = 0; }
    }
  }
  if (a.frontier_bits) {
    const long long nwords = (n + 31) / 32;
    for (long long wd = tid; wd < nwords; wd += nth) a.frontier_bits[wd] = (wd == (src >> 5)) ? 1u << (src & 31) : 0u;
  }
  if (tid == 0) {
    bfs_ctrl_reset(a.ctrl);
    const u32 ro = a.row_offsets[src];
    const u32 deg = a.row_offsets[src + 1] - ro;
    a.q_row[0][0] = ro;
    a.q_off[0][0] = 0;
    a.q_du[0][0] = (u32)src;
    a.ctrl->cursor[0] = deg ? ((1ull << BFS_VSHIFT) | (u64)deg) : 0ull;
    a.ctrl->sssp_thr = a.delta > 0.f ? __float_as_uint(a.delta) : SSSP_INF_BITS;     // bucket 0: [0, delta)
  }
}

// bookkeeping of iteration `it` (one thread; rides on the relax launch: nothing it writes is read by that launch)
__device__ __forceinline__ void sssp_open(const sssp_args_t& a, int it) {
  bfs_ctrl_t* const c = a.ctrl;
  const u64 cur = c->cursor[it % 3];
  c->cursor[(it + 2) % 3] = 0;
  if (it < 64) c->stamp[it] = __builtin_amdgcn_s_memrealtime();
  // what the previous build left waiting behind the threshold; this iteration's build counts into the other pair
  const u32 far_cnt = c->sssp_far_cnt[it & 1], far_min = c->sssp_far_min[it & 1];
  c->sssp_far_cnt[(it + 1) & 1] = 0;
  c->sssp_far_min[(it + 1) & 1] = SSSP_INF_BITS;
  if ((cur >> BFS_VSHIFT) == 0) {
    if (far_cnt && a.delta > 0.f) {
      // the near queue ran dry: next bucket = the one that holds the smallest waiting distance (this iteration's
      // relax finds nothing to do, its build sweeps the marks again with the new threshold)
      const float lo = floorf(__uint_as_float(far_min) / a.delta);
      u32 thr = __float_as_uint((lo + 1.0f) * a.delta);
      // progress guarantee: in float32 (lo + 1) * delta can round to <= far_min once far_min / delta nears 2^23 (far_min
      // 1000, delta 1e-5: thr == 1000) -- the build tests d >= thr, nothing would become near and the same threshold
      // would be computed forever.  Non-negative floats order like their bit patterns: the next float above far_min
      // always admits the smallest waiting vertex.
      if (thr <= far_min) thr = far_min + 1u;
      c->sssp_thr = thr;
      return;
    }
    if (!c->done) { c->done = 1; c->levels = it; }
    return;
  }
  if (it < BFS_MAX_TRACE) c->trace[it] = cur;
  c->sum_edges += cur & BFS_EMASK;            // relaxations
  c->sum_frontier += cur >> BFS_VSHIFT;
}

constexpr int SSSP_EPT = 4;
constexpr int SSSP_TILE = WAVE * SSSP_EPT;
// Upper bounds of the distances of the first SSSP_HOTN vertices (the hubs under the hub-first layout: the targets
// of a large share of the edges), 16 bits each, in LDS: a candidate that is not below the bound cannot improve the
// distance and skips the gather from the 16 MB distance array -- what bounds this kernel (80 G relaxations/s is the
// rate of 4-byte gathers from a table of that size, profiles/r01/microbench.jsonl).  A bound is the distance at the
// time the workgroup started, rounded UP to bfloat16: distances only decrease, so it stays a bound.
constexpr int SSSP_HOTN = 32768;                   // 64 KB per workgroup, two workgroups per CU (65536 with one: same rate)
constexpr u32 SSSP_HOT_MIN_EDGES = 1u << 20;       // smaller iterations do not pay for the copy

// ---- heavy iterations: a sweep over the unit blocks and the degree classes instead of the queue ------------------------
// The queue walk above resolves every edge rank to its row (a search per tile) and reads neighbours and weights 4 bytes per
// lane.  An iteration whose frontier holds a large share of ALL edges (the three heavy iterations of an RMAT-22 run relax
// 80-100 M of the 134 M edges each: 2.8 of the run's 3.5 ms) is cheaper as a sweep in the shape of the BFS's unit-block
// body and of mgx/nreduce.hpp: the long rows' entries and weights are streamed from the unit blocks, 16 bytes per lane, a
// unit of 64 entries belongs to ONE row -- its owner's frontier bit and current distance are one (broadcast) gather per
// unit --, the short rows are walked by degree class; no queue, no search.  What stays random is dist[dst]: the hubs'
// bounds sit in LDS as before.  Rows outside the frontier are read and masked: the price of the sweep.
typedef unsigned int sssp_u32x4 __attribute__((ext_vector_type(4)));
typedef float sssp_f32x4 __attribute__((ext_vector_type(4)));
struct __attribute__((aligned(4))) sssp_u32x4u { u32 x, y, z, w; };
struct __attribute__((aligned(4))) sssp_f32x4u { float x, y, z, w; };

// one candidate: entry d (0xFFFFFFFF: none) reached with distance bits nd
__device__ __forceinline__ u32 sssp_gather_index(u32 d, u32 nd, const unsigned short* hot16, u32 hot_n, bool& live) {
  const bool none = d == 0xFFFFFFFFu;
  const bool hotm = d < hot_n;
  const u32 ub = hot16[hotm ? d : 0u];
  const bool skip = hotm && nd >= (ub << 16);          // cannot beat the bound of a hub: no gather
  live = !none && !skip;
  return live ? d : 0u;
}

// four candidates of a lane.  The table holds 16-bit upper bounds of the first hot_n distances, taken when the workgroup started: a
// candidate below the bound gathers the distance.  (32-bit minima the workgroup keeps current with ds_min instead -- MGX_SSSP_LIVE of the lab
// builds until round 6 -- measured 2.56 against 1.88 ms per RMAT-22 source: half the vertices fit, and every hot candidate is an LDS atomic.)
__device__ __forceinline__ void sssp_relax4(const u32 (&dd)[4], const u32 (&nd)[4], const void* table, u32 hot_n, u32* dist, unsigned char* mark) {
  u32 gi[4], old[4];
  bool live[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    gi[k] = sssp_gather_index(dd[k], nd[k], (const unsigned short*)table, hot_n, live[k]);
  }
  // (the gathers stay unconditional: a plain load whose only use sits behind live[k] is sunk under it by the code generator, every gather
  //  then a round trip of its own behind an s_waitcnt vmcnt(0) inside the branch -- mgx/nreduce.hpp: nr_load_pinned.  The comparison
  //  is made for every candidate, outside the branch.)
#pragma unroll
  for (int k = 0; k < 4; ++k) old[k] = dist[gi[k]];
  bool go[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) go[k] = (nd[k] < old[k]) & live[k];
#pragma unroll
  for (int k = 0; k < 4; ++k)
    if (go[k]) {
      atomicMin(dist + dd[k], nd[k]);
      mark[dd[k]] = 1;
    }
}

typedef unsigned int sssp_u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned int sssp_u32x2 __attribute__((ext_vector_type(2)));

// PACKED: the entries as 24-bit ids (12 bytes per lane and load) and the weights as halves (8 bytes) -- a.ub_col24 / a.ub_w16.
// A unit's padding entries are -1 (no candidate: sssp_gather_index), a padding unit belongs to vertex n: nothing has to say
// how many entries of a unit are real.
// PIDS / PW separately (round 6): a layout that carries the 24-bit copy has dropped the 32-bit entries, whatever the weights are -- real
// weights then ride as floats beside 24-bit ids (7 bytes per entry).
template <int NT, bool PIDS, bool PW>
__device__ __forceinline__ void sssp_dense_long(const sssp_args_t& a, void* table, u32 hot_n, u32 block, u32 nblocks) {
  constexpr int NW = NT / WAVE;
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
  const u32 sub = (u32)lane & 15u, q = (u32)lane >> 4;
  const u32 H = a.ub_units_pad / 8u;                   // half-groups: 8 units = 512 entries
  const u32 W = nblocks * NW, w = block * NW + (u32)wave;
  if (w >= H) return;
  const int* __restrict__ ucol = a.ub_col;
  const float* __restrict__ uw = a.ub_w;
  const u32* __restrict__ ucol24 = a.ub_col24;
  const unsigned short* __restrict__ uw16 = a.ub_w16;
  const int* __restrict__ owner = a.ub_owner;
  const u32* __restrict__ fbits = a.frontier_bits;
  u32* dist = a.dist;
  unsigned char* mark = a.mark;
  typedef typename std::conditional<PIDS, sssp_u32x3, sssp_u32x4>::type craw_t;
  typedef typename std::conditional<PW, sssp_u32x2, sssp_f32x4>::type wraw_t;
  // GATED (round 6): a unit whose row is not in the frontier is never streamed.  The sweep used to read every long-row entry and
  // weight -- 5 bytes packed, 610 MB of unit blocks on RMAT-22 -- and mask the inactive ones afterwards: 456 MB per dispatch (PMC)
  // whatever the frontier held.  Now, as the BFS's unit-block body does it (bfs_fused_dense.hpp), the owners run ahead of the stream:
  //   stage A  owners of half-group h + 3 W                       (8 bytes per 8 units and wave: coalesced)
  //   stage B  frontier words of the owners of h + 2 W            (a gather per unit, 16 lanes the same word)
  //   stage C  h + W: entries and weights of the ACTIVE units (lanes of the others read the zero line behind the blocks: cached,
  //            no HBM traffic), and the rows' distances NOW
  //   stage D  h: relax -- skipped as a whole when the wave holds no active unit (no LDS look, no gather, no round trip)
  // A row's distance is read one stage earlier than before (with its entries, not right before the relaxation): still a value
  // the row held during this iteration, and a row that improves later is marked for the next one -- same fixed point.
  craw_t cC[2], cN[2];
  wraw_t wC[2], wN[2];
  u32 duC[2], duN[2];
  bool aC[2], aN[2];
  const u32 dummy = a.ub_units_pad << 6;                // entry index of the padding behind the blocks (4 x -1 / four zero weights)
  auto load_owner = [&](u32 h, u32* own) {
    const u32 hh = h < H ? h : H - 1u;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const u32 o = (u32)owner[hh * 8u + 4u * (u32)j + q];
      own[j] = h < H ? o : (u32)a.n;                   // (past the end: a padding unit; padding units belong to vertex n)
    }
  };
  auto gather_fw = [&](const u32* own, u32* fw) {
#pragma unroll
    for (int j = 0; j < 2; ++j) fw[j] = fbits[(own[j] < (u32)a.n ? own[j] : 0u) >> 5];
  };
  auto issue = [&](u32 h, const u32* own, const u32* fw, craw_t* c, wraw_t* wt, u32* du, bool* act) {
    const u32 hh = h < H ? h : H - 1u;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      act[j] = own[j] < (u32)a.n && ((fw[j] >> (own[j] & 31u)) & 1u);
      const u32 u = hh * 8u + 4u * (u32)j + q;
      const size_t e = act[j] ? ((size_t)u << 6) + sub * 4u : (size_t)dummy;
      if constexpr (PIDS) c[j] = __builtin_nontemporal_load((const sssp_u32x3*)(ucol24 + (e >> 2) * 3u));
      else c[j] = __builtin_nontemporal_load((const sssp_u32x4*)(ucol + e));
      if constexpr (PW) wt[j] = __builtin_nontemporal_load((const sssp_u32x2*)(uw16 + e));
      else wt[j] = __builtin_nontemporal_load((const sssp_f32x4*)(uw + e));
      du[j] = dist[act[j] ? own[j] : 0u];
    }
  };
  u32 o1[2], o2[2], o3[2], f1[2], f2[2];
  load_owner(w, o1); load_owner(w + W, o2); load_owner(w + 2u * W, o3);
  gather_fw(o1, f1);
  gather_fw(o2, f2);
  issue(w, o1, f1, cC, wC, duC, aC);
  for (u32 h = w; h < H; h += W) {
    u32 o4[2], f3[2];
    load_owner(h + 3u * W, o4);                        // A
    gather_fw(o3, f3);                                 // B
    issue(h + W, o2, f2, cN, wN, duN, aN);             // C
    if (__ballot(aC[0] || aC[1])) {                    // D (wave-uniform)
#pragma unroll
      for (int j = 0; j < 2; ++j) {                    // (one unit load at a time: 64 registers per lane)
        const float base = __uint_as_float(duC[j]);
        u32 e4[4];
        float w4[4];
        if constexpr (PIDS) {
          e4[0] = (u32)__builtin_amdgcn_sbfe((int)cC[j].x, 0, 24);
          e4[1] = (u32)__builtin_amdgcn_sbfe((int)__builtin_amdgcn_alignbit(cC[j].y, cC[j].x, 24), 0, 24);
          e4[2] = (u32)__builtin_amdgcn_sbfe((int)__builtin_amdgcn_alignbit(cC[j].z, cC[j].y, 16), 0, 24);
          e4[3] = (u32)((int)cC[j].z >> 8);
        } else {
          e4[0] = cC[j].x; e4[1] = cC[j].y; e4[2] = cC[j].z; e4[3] = cC[j].w;
        }
        if constexpr (PW) {
          w4[0] = __half2float(__ushort_as_half((unsigned short)(wC[j].x & 0xFFFFu))); w4[1] = __half2float(__ushort_as_half((unsigned short)(wC[j].x >> 16)));
          w4[2] = __half2float(__ushort_as_half((unsigned short)(wC[j].y & 0xFFFFu))); w4[3] = __half2float(__ushort_as_half((unsigned short)(wC[j].y >> 16)));
        } else {
          w4[0] = wC[j].x; w4[1] = wC[j].y; w4[2] = wC[j].z; w4[3] = wC[j].w;
        }
        u32 dd[4], nd[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          dd[k] = aC[j] ? e4[k] : 0xFFFFFFFFu;         // (-1 padding entries stay -1)
          nd[k] = __float_as_uint(base + w4[k]);
        }
        sssp_relax4(dd, nd, table, hot_n, dist, mark);
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      cC[j] = cN[j]; wC[j] = wN[j]; duC[j] = duN[j]; aC[j] = aN[j];
      o2[j] = o3[j]; f2[j] = f3[j]; o3[j] = o4[j];
    }
  }
}

template <int NT>
__device__ __forceinline__ void sssp_dense_short(const sssp_args_t& a, void* table, u32 hot_n, u32 block, u32 nblocks) {
  constexpr int NW = NT / WAVE;
  const int lane = lane_id();
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
  const u32 b0 = a.vs_v[0], b1 = a.vs_v[1], b2 = a.vs_v[2], b3 = a.vs_v[3];
  const u32 s16 = (b1 - b0 + 3u) / 4u, s4 = (b2 - b1 + 15u) / 16u, s1 = (b3 - b2 + 63u) / 64u;
  const u32 T = s16 + s4 + s1;
  const u32 W = nblocks * NW, w = block * NW + (u32)wave;
  const u32* __restrict__ ro = a.row_offsets;
  const int* __restrict__ col = a.col_indices;
  const float* __restrict__ wts = a.weights;
  const u32* __restrict__ fbits = a.frontier_bits;
  u32* dist = a.dist;
  unsigned char* mark = a.mark;
  struct plan_t { u32 e0, cnt, du; };
  auto plan = [&](u32 s) -> plan_t {
    plan_t p; p.e0 = 0u; p.cnt = 0u; p.du = SSSP_INF_BITS;
    u32 lpr_shift, vbase, vend;
    if (s < s16) { lpr_shift = 4; vbase = b0 + s * 4u; vend = b1; }
    else if (s < s16 + s4) { lpr_shift = 2; vbase = b1 + (s - s16) * 16u; vend = b2; }
    else { lpr_shift = 0; vbase = b2 + (s - s16 - s4) * 64u; vend = b3; }
    const u32 v = vbase + ((u32)lane >> lpr_shift);
    const u32 sub = (u32)lane & ((1u << lpr_shift) - 1u);
    const bool in = s < T && v < vend;
    const u32 vc = in ? v : 0u;
    const u32 lo = ro[vc], hi = ro[vc + 1];
    const u32 fw = fbits[vc >> 5];
    p.du = dist[vc];
    const u32 deg = hi - lo;
    if (in && ((fw >> (vc & 31u)) & 1u) && sub * 4u < deg) { p.e0 = lo + sub * 4u; p.cnt = deg - sub * 4u < 4u ? deg - sub * 4u : 4u; }
    return p;
  };
  if (w >= T) return;
  plan_t pc = plan(w);
  sssp_u32x4u dc = *(const sssp_u32x4u*)(col + pc.e0);
  sssp_f32x4u wc = *(const sssp_f32x4u*)(wts + pc.e0);
  for (u32 s = w; s < T; s += W) {
    const plan_t pn = plan(s + W);
    const sssp_u32x4u dn = *(const sssp_u32x4u*)(col + pn.e0);
    const sssp_f32x4u wn = *(const sssp_f32x4u*)(wts + pn.e0);
    const float base = __uint_as_float(pc.du);
    u32 dd[4], nd[4];
    dd[0] = pc.cnt > 0u ? dc.x : 0xFFFFFFFFu; dd[1] = pc.cnt > 1u ? dc.y : 0xFFFFFFFFu;
    dd[2] = pc.cnt > 2u ? dc.z : 0xFFFFFFFFu; dd[3] = pc.cnt > 3u ? dc.w : 0xFFFFFFFFu;
    nd[0] = __float_as_uint(base + wc.x); nd[1] = __float_as_uint(base + wc.y);
    nd[2] = __float_as_uint(base + wc.z); nd[3] = __float_as_uint(base + wc.w);
    sssp_relax4(dd, nd, table, hot_n, dist, mark);
    pc = pn; dc = dn; wc = wn;
  }
}

// weights of the unit blocks: ub_w[(ub_first[v] << 6) + k] = weights[row_offsets[v] + k] for the rows that hold units
__global__ __launch_bounds__(BLOCK) void k_sssp_unit_weights(const int* __restrict__ ro, const float* __restrict__ wts,
                                                             const int* __restrict__ ub_first, int n, float* __restrict__ ub_w) {
  const int lane = threadIdx.x & 63;
  const long long wave0 = ((long long)blockIdx.x * BLOCK + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * BLOCK) >> 6;
  for (long long v = wave0; v < n; v += nwaves) {
    const int u0 = ub_first[v], u1 = ub_first[v + 1];
    if (u1 == u0) continue;
    const int r0 = ro[v], deg = ro[v + 1] - r0;
    const long long e0 = (long long)u0 << 6, e1 = (long long)u1 << 6;
    for (long long e = e0 + lane; e < e1; e += 64) ub_w[e] = (e - e0) < deg ? wts[r0 + (e - e0)] : 0.0f;
  }
}


// the unit blocks' weights as halves -- *exact = 0 if any weight is not the float its half converts back to
__global__ __launch_bounds__(BLOCK) void k_sssp_unit_weights16(const float* __restrict__ ub_w, long long entries, unsigned short* __restrict__ ub_w16,
                                                               int* exact) {
  bool bad = false;
  for (long long e = (long long)blockIdx.x * BLOCK + threadIdx.x; e < entries; e += (long long)gridDim.x * BLOCK) {
    const float w = ub_w[e];
    const __half h = __float2half(w);
    bad |= __half2float(h) != w;
    ub_w16[e] = __half_as_ushort(h);
  }
  if (__ballot(bad) && lane_id() == 0) *exact = 0;
}

// (A destination-sliced list of ALL edges as (src, dst, w) triples with a slice of the distances in LDS was the option MGX_SSSP_SLICED
// until round 6: distances identical, the first heavy RMAT-22 iteration 0.89 ms that way too -- LDS minima of a slice full of hubs collide,
// and what a workgroup finds reaches the array only at the end of its piece, so the iteration propagates less: 456 M relaxations
// instead of 346 M.  Lost twice (rounds 2 and 5), removed; the code is in the history of this file and of mgx_layout.hip.)

// bounds of the first HOTN distances into LDS (two per word); returns how many there are.  All loads first, then the stores
// (the plain loop compiles to load - wait - store per trip: sixteen dependent round trips at the start of every workgroup;
// bfs_copy_prefix in bfs_fused.hpp tells the same story)
template <int NT, int HOTN>
__device__ __forceinline__ u32 sssp_load_bounds(const u32* __restrict__ dist, int n, u32* s_hot) {
  const u32 hot_n = (u32)n < (u32)HOTN ? ((u32)n & ~1u) : (u32)HOTN;
  constexpr int IT = (HOTN / 2 + NT - 1) / NT;
  uint2 dv[IT];
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const u32 i = (u32)k * NT + threadIdx.x;
    dv[k] = *(const uint2*)(dist + 2 * (i < hot_n / 2 ? i : 0u));
  }
#pragma unroll
  for (int k = 0; k < IT; ++k) {
    const u32 i = (u32)k * NT + threadIdx.x;
    if (i < hot_n / 2) s_hot[i] = ((dv[k].x + 0xFFFFu) >> 16) | (((dv[k].y + 0xFFFFu) >> 16) << 16);
  }
  __syncthreads();
  return hot_n;
}

// The heavy iterations' kernel (sssp_dense_*): launched behind every k_sssp_relax of the loop and returns at once unless the
// iteration is heavy -- then k_sssp_relax did (the same grid-uniform test on the same stable sizes).  A launch of its own
// for the shape that suits a sweep whose cost is its distance gathers (mgx/nreduce.hpp measured the same trade): ONE
// workgroup per CU, 128 registers per lane, and the bounds of the first SSSP_HOTN_DENSE vertices in LDS.
constexpr int SSSP_HOTN_DENSE = 73728;             // 144 KB of 16-bit bounds
template <int NT>
__global__ __launch_bounds__(NT, 4) void k_sssp_relax_dense(sssp_args_t a, int it) {
  extern __shared__ __attribute__((aligned(16))) u32 s_hot_dense[];
  const u64 cur = a.ctrl->cursor[it % 3];
  const u32 E = (u32)(cur & BFS_EMASK);
  if ((cur >> BFS_VSHIFT) == 0 || !a.ub_w || (u64)E * (u64)a.dense_div < a.m_edges) return;
  const u32 hot_n = sssp_load_bounds<NT, SSSP_HOTN_DENSE>(a.dist, a.n, s_hot_dense);
  if (a.ub_col24 && a.ub_w16) sssp_dense_long<NT, true, true>(a, s_hot_dense, hot_n, blockIdx.x, gridDim.x);      // (grid-uniform)
  else if (a.ub_col24) sssp_dense_long<NT, true, false>(a, s_hot_dense, hot_n, blockIdx.x, gridDim.x);
  else sssp_dense_long<NT, false, false>(a, s_hot_dense, hot_n, blockIdx.x, gridDim.x);
  sssp_dense_short<NT>(a, s_hot_dense, hot_n, blockIdx.x, gridDim.x);
}

template <int NT>
__global__ __launch_bounds__(NT, 8) void k_sssp_relax(sssp_args_t a, int it) {
  constexpr int NW = NT / WAVE;
  constexpr int EPT = SSSP_EPT;
  __shared__ u32 s_off[NW][68];
  __shared__ u32 s_row[NW][64];
  __shared__ u32 s_du[NW][64];
  const int wave = threadIdx.x / WAVE;
  const int lane = lane_id();
  u32* const w_off = s_off[wave];
  u32* const w_row = s_row[wave];
  u32* const w_du = s_du[wave];

  bfs_ctrl_t* const c = a.ctrl;
  const u64 cur = c->cursor[it % 3];
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) sssp_open(a, it);
  const long long nf = (long long)(cur >> BFS_VSHIFT);
  const u32 E = (u32)(cur & BFS_EMASK);
  if (nf == 0) return;
  const u32* __restrict__ fr_row = a.q_row[it & 1];
  const u32* __restrict__ fr_off = a.q_off[it & 1];
  const u32* __restrict__ fr_du = a.q_du[it & 1];
  const int* __restrict__ col = a.col_indices;
  const float* __restrict__ wts = a.weights;
  u32* dist = a.dist;
  unsigned char* mark = a.mark;

  extern __shared__ __attribute__((aligned(16))) u32 s_hot[];        // SSSP_HOTN / 2 words: two bounds per word
  if (a.ub_w && (u64)E * (u64)a.dense_div >= a.m_edges) return;     // a heavy iteration (grid-uniform): k_sssp_relax_dense's
  const bool use_hot = E >= a.hot_min_edges;
  const u32 hot_n = use_hot ? sssp_load_bounds<NT, SSSP_HOTN>(dist, a.n, s_hot) : 0u;
  const unsigned short* const hot16 = (const unsigned short*)s_hot;

  const u32 total_waves = gridDim.x * NW;
  u32 per = (E + total_waves - 1) / total_waves;
  per = (per + WAVE - 1) / WAVE * WAVE;
  const u64 rb = (u64)(blockIdx.x * NW + wave) * per;
  const bool has_work = rb < (u64)E;
  const u32 r_begin = has_work ? (u32)rb : E;
  const u32 r_end = (rb + per < (u64)E) ? (u32)(rb + per) : E;
  if (has_work) {

  long long seg = wave_upper_bound(fr_off, nf, r_begin) - 1;
  u32 pf_off = 0, pf_row = 0, pf_du = 0, pf_off_last = 0;
  bool pf_ok = false, pf_last_ok = false;
  auto prefetch = [&](long long sg) {
    const long long s0 = sg + lane;
    const long long s1 = sg + WAVE;
    pf_ok = s0 < nf;
    pf_last_ok = s1 < nf;
    pf_off = fr_off[pf_ok ? s0 : nf - 1];
    pf_row = fr_row[pf_ok ? s0 : nf - 1];
    pf_du = fr_du[pf_ok ? s0 : nf - 1];
    pf_off_last = fr_off[pf_last_ok ? s1 : nf - 1];
  };
  prefetch(seg);

  u32 eidxC[EPT], duC[EPT];
  u32 actC = 0;
  u32 e_next = r_begin;
  auto prepare_tile = [&]() {
    const u32 E0 = e_next;
    const u32 my = pf_ok ? pf_off : E;
    w_off[lane] = my;
    w_row[lane] = pf_ok ? pf_row : 0u;
    w_du[lane] = dist[pf_du];                          // pf_du = vertex id of the row (valid id also past the queue end)
    const u32 off64 = pf_last_ok ? pf_off_last : E;
    if (lane == 0) w_off[WAVE] = off64;
    wave_lds_fence();          // the reads below are of OTHER lanes' slots
    u32 E1 = (r_end - E0 > (u32)SSSP_TILE) ? E0 + SSSP_TILE : r_end;
    if (off64 < E1) E1 = off64;
    const u32 nxt = w_off[lane + 1];
    const u64 m = __ballot(my < E1 && nxt >= E1);
    const int nseg = __ffsll((long long)m);
    const u32 next_off = w_off[nseg];
    const long long seg_next = seg + ((next_off == E1) ? nseg : nseg - 1);
    prefetch(seg_next);
    u32 r[EPT];
    int sj[EPT];
    actC = 0;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const u32 rr = E0 + (u32)(k * WAVE + lane);
      const bool act = rr < E1;
      if (act) actC |= 1u << k;
      r[k] = act ? rr : E0;
      sj[k] = 0;
    }
    const u32 off0 = w_off[0];
    const u32 d0 = w_off[1] - off0;
    if (nseg > 1) {
      int top = 1;
      while (top * 2 < nseg) top *= 2;
      for (int step = top; step > 0; step >>= 1) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
          const int j = sj[k] + step;
          const u32 vv = w_off[j < nseg ? j : nseg - 1];
          if (j < nseg && vv <= r[k]) sj[k] = j;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      eidxC[k] = w_row[sj[k]] + (r[k] - w_off[sj[k]]);
      duC[k] = w_du[sj[k]];
    }
    seg = seg_next;
    e_next = E1;
  };

  // three stages, all loads unconditional: tile t loads (neighbour, weight), tile t-1 gathers the neighbours'
  // distances, tile t-2 is tested
  int vA[EPT], vB[EPT];
  u32 ndA[EPT], oldA[EPT], duB[EPT];
  float wB[EPT];
  u32 actA = 0, actB = 0;
#pragma unroll
  for (int k = 0; k < EPT; ++k) { vA[k] = 0; vB[k] = 0; ndA[k] = SSSP_INF_BITS; oldA[k] = 0; duB[k] = 0; wB[k] = 0.f; }
  bool haveA = false, haveB = false, haveC = true;
  prepare_tile();
  while (haveA || haveB || haveC) {
    if (haveA) {
#pragma unroll
      for (int k = 0; k < EPT; ++k) {
        if (((actA >> k) & 1u) && ndA[k] < oldA[k]) {
          atomicMin(dist + vA[k], ndA[k]);
          mark[vA[k]] = 1;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      vA[k] = vB[k];
      ndA[k] = __float_as_uint(__uint_as_float(duB[k]) + wB[k]);
    }
    actA = haveB ? actB : 0u;
    // candidates that cannot beat the bound of a hot neighbour: no gather (the lane reads dist[0] with the others)
    u32 gA[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const bool hotm = (u32)vA[k] < hot_n;
      const u32 ub = hot16[hotm ? (u32)vA[k] : 0u];
      const bool skip = hotm && ndA[k] >= (ub << 16);
      if (skip) actA &= ~(1u << k);
      gA[k] = skip ? 0u : (u32)vA[k];
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      u32 ei = haveC ? eidxC[k] : 0u;
      vB[k] = col[ei];
      wB[k] = wts[ei];
      duB[k] = duC[k];
    }
    actB = haveC ? actC : 0u;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      oldA[k] = dist[gA[k]];
    }
    haveA = haveB;
    haveB = haveC;
    haveC = e_next < r_end;
    if (haveC) prepare_tile();
  }
  }
}

// marked vertices -> next frontier; marks cleared.  16 vertices per thread, see k_bfs_build.
template <int NT>
__global__ __launch_bounds__(NT) void k_sssp_build(sssp_args_t a, int it) {
  constexpr int NW = NT / WAVE;
  constexpr int LIST = 8 * NT;
  constexpr int PER = LIST / NT;
  constexpr u64 CNT1 = 1ull << 40;
  constexpr u64 DEGMASK = CNT1 - 1ull;
  __shared__ u32 st_v[LIST];
  __shared__ u64 s_scan[NW + 1];
  __shared__ u64 s_base;
  bfs_ctrl_t* const c = a.ctrl;
  if (c->done) return;
  const long long i0 = (((long long)blockIdx.x + (long long)(threadIdx.x >> 6) * gridDim.x) * 64 + (threadIdx.x & 63)) * 16;
  u32 new16 = 0;
  u32 far_n = 0, far_lo = SSSP_INF_BITS;
  if (i0 < a.n) {
    const u32 valid = (a.n - i0 >= 16) ? 0xFFFFu : ((1u << (int)(a.n - i0)) - 1u);
    uint4* mp = (uint4*)(a.mark + i0);
    const uint4 m = *mp;
    if (m.x | m.y | m.z | m.w) {
      const u32 x[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) new16 |= (((x[q] & 0x01010101u) * 0x10204080u) >> 28) << (4 * q);
      new16 &= valid;
      uint4 keep = make_uint4(0, 0, 0, 0);
      if (a.delta > 0.f) {
        // near / far: only distances below the threshold enter the queue now; the others keep their mark
        const u32 thr = c->sssp_thr;
        u32 far16 = 0;
        for (u32 rest = new16; rest;) {
          const int q = __ffs((int)rest) - 1;
          rest &= rest - 1;
          const u32 d = a.dist[i0 + q];
          if (d >= thr) { far16 |= 1u << q; far_lo = d < far_lo ? d : far_lo; }
        }
        new16 &= ~far16;
        far_n = (u32)__popc(far16);
        u32 kb[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 16; ++q) kb[q >> 2] |= ((far16 >> q) & 1u) << (8 * (q & 3));
        keep = make_uint4(kb[0], kb[1], kb[2], kb[3]);
      }
      *mp = keep;
    }
  }
  if (a.delta > 0.f) {               // (grid-uniform) what stays behind: count and smallest distance, one atomic pair per wave
    const u32 wn = wave_sum(far_n);
    u32 wl = far_lo;
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) { const u32 o = (u32)__shfl_xor((int)wl, d, WAVE); wl = o < wl ? o : wl; }
    if ((threadIdx.x & (WAVE - 1)) == 0 && wn) {
      atomicAdd(&c->sssp_far_cnt[(it + 1) & 1], wn);
      atomicMin(&c->sssp_far_min[(it + 1) & 1], wl);
    }
  }
  u64 total64;
  const u64 before = block_exclusive_sum_nw<NW>((u64)__popc(new16), s_scan, &total64);
  const int total = (int)total64;
  if (total == 0) return;
  u64* const cur = &c->cursor[(it + 1) % 3];
  u32* __restrict__ const out_row = a.q_row[(it + 1) & 1];
  u32* __restrict__ const out_off = a.q_off[(it + 1) & 1];
  u32* __restrict__ const out_du = a.q_du[(it + 1) & 1];
  for (int first = 0; first < total; first += LIST) {
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
    __syncthreads();
    const int cnt = (total - first < LIST) ? total - first : LIST;
    u32 ro[PER], ro1[PER], du[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = threadIdx.x * PER + q;
      const u32 v = (i < cnt) ? st_v[i] : 0u;
      const bfs_u32x2 ext = *(const bfs_u32x2*)(a.row_offsets + v);
      ro[q] = ext.x;
      ro1[q] = ext.y;
      du[q] = v;
    }
    u64 loc[PER];
    u64 sum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = threadIdx.x * PER + q;
      const u32 deg = (i < cnt) ? ro1[q] - ro[q] : 0u;
      loc[q] = sum;
      sum += deg ? (CNT1 | (u64)deg) : 0ull;
    }
    u64 tot;
    const u64 ex = block_exclusive_sum_nw<NW>(sum, s_scan, &tot);
    if (threadIdx.x == 0) s_base = (tot >> 40) ? atomicAdd(cur, ((tot >> 40) << BFS_VSHIFT) | (tot & DEGMASK)) : 0ull;
    __syncthreads();
    const u64 base = s_base;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int i = threadIdx.x * PER + q;
      if (i < cnt && ro1[q] != ro[q]) {
        const u64 at = ex + loc[q];
        const u64 slot = (base >> BFS_VSHIFT) + (at >> 40);
        out_row[slot] = ro[q];
        out_off[slot] = (u32)((base & BFS_EMASK) + (at & DEGMASK));
        out_du[slot] = du[q];
      }
    }
    __syncthreads();
  }
}

// The same without the detour through a list (k_bfs_build2's shape): a thread's 16 vertices are consecutive, so their row
// extents are ONE contiguous piece (17 offsets, 16-byte loads) read by every thread that found a mark; one packed
// workgroup scan gives the positions, a thread writes its own queue entries.  Three barriers instead of the list's
// redistribution and its two scans per batch: 28 -> ~10 us for the light iterations at the end of an RMAT-22 run.
// Needs row_offsets 16-byte aligned (the host checks).
template <int NT>
__global__ __launch_bounds__(NT, 4) void k_sssp_build2(sssp_args_t a, int it) {
  constexpr int NW = NT / WAVE;
  constexpr u64 CNT1 = 1ull << 40;
  constexpr u64 DEGMASK = CNT1 - 1ull;
  __shared__ u64 s_scan[NW + 1];
  __shared__ u64 s_base;
  bfs_ctrl_t* const c = a.ctrl;
  if (c->done) return;
  const long long i0 = (((long long)blockIdx.x + (long long)(threadIdx.x >> 6) * gridDim.x) * 64 + (threadIdx.x & 63)) * 16;
  u32 new16 = 0;
  u32 far_n = 0, far_lo = SSSP_INF_BITS;
  if (i0 < a.n) {
    const u32 valid = (a.n - i0 >= 16) ? 0xFFFFu : ((1u << (int)(a.n - i0)) - 1u);
    uint4* mp = (uint4*)(a.mark + i0);
    const uint4 m = *mp;
    if (m.x | m.y | m.z | m.w) {
      const u32 x[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
      for (int q = 0; q < 4; ++q) new16 |= (((x[q] & 0x01010101u) * 0x10204080u) >> 28) << (4 * q);
      new16 &= valid;
      uint4 keep = make_uint4(0, 0, 0, 0);
      if (a.delta > 0.f) {
        // near / far: only distances below the threshold enter the queue now; the others keep their mark
        const u32 thr = c->sssp_thr;
        u32 far16 = 0;
        for (u32 rest = new16; rest;) {
          const int q = __ffs((int)rest) - 1;
          rest &= rest - 1;
          const u32 d = a.dist[i0 + q];
          if (d >= thr) { far16 |= 1u << q; far_lo = d < far_lo ? d : far_lo; }
        }
        new16 &= ~far16;
        far_n = (u32)__popc(far16);
        u32 kb[4] = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 16; ++q) kb[q >> 2] |= ((far16 >> q) & 1u) << (8 * (q & 3));
        keep = make_uint4(kb[0], kb[1], kb[2], kb[3]);
      }
      *mp = keep;
    }
    if (a.frontier_bits) ((unsigned short*)a.frontier_bits)[i0 >> 4] = (unsigned short)new16;    // the next frontier as a bitmap (the sweep: sssp_dense_*)
  }
  if (a.delta > 0.f) {               // (grid-uniform) what stays behind: count and smallest distance, one atomic pair per wave
    const u32 wn = wave_sum(far_n);
    u32 wl = far_lo;
#pragma unroll
    for (int d = WAVE / 2; d > 0; d >>= 1) { const u32 o = (u32)__shfl_xor((int)wl, d, WAVE); wl = o < wl ? o : wl; }
    if ((threadIdx.x & (WAVE - 1)) == 0 && wn) {
      atomicAdd(&c->sssp_far_cnt[(it + 1) & 1], wn);
      atomicMin(&c->sssp_far_min[(it + 1) & 1], wl);
    }
  }
  // my vertices' row extents: contiguous
  u32 ro[17];
  u64 sum = 0;
  if (new16) {
    if (i0 + 16 < (long long)a.n) {
      const u32* const p = a.row_offsets + i0;
      const uint4 r0 = *(const uint4*)p, r1 = *(const uint4*)(p + 4), r2 = *(const uint4*)(p + 8), r3 = *(const uint4*)(p + 12);
      ro[0] = r0.x; ro[1] = r0.y; ro[2] = r0.z; ro[3] = r0.w; ro[4] = r1.x; ro[5] = r1.y; ro[6] = r1.z; ro[7] = r1.w;
      ro[8] = r2.x; ro[9] = r2.y; ro[10] = r2.z; ro[11] = r2.w; ro[12] = r3.x; ro[13] = r3.y; ro[14] = r3.z; ro[15] = r3.w;
      ro[16] = p[16];
    } else {
#pragma unroll
      for (int q = 0; q < 17; ++q) ro[q] = a.row_offsets[(i0 + q <= (long long)a.n) ? i0 + q : (long long)a.n];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const u32 deg = ro[q + 1] - ro[q];
      if (((new16 >> q) & 1u) && deg) sum += CNT1 | (u64)deg;
    }
  }
  u64 tot;
  u64 ex = block_exclusive_sum_lean<NW>(sum, s_scan, &tot);
  if ((tot >> 40) == 0) return;
  if (threadIdx.x == 0) s_base = atomicAdd(&c->cursor[(it + 1) % 3], ((tot >> 40) << BFS_VSHIFT) | (tot & DEGMASK));
  __syncthreads();
  if (!new16) return;
  const u64 base = s_base;
  u32* __restrict__ const out_row = a.q_row[(it + 1) & 1];
  u32* __restrict__ const out_off = a.q_off[(it + 1) & 1];
  u32* __restrict__ const out_du = a.q_du[(it + 1) & 1];
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const u32 deg = ro[q + 1] - ro[q];
    if (((new16 >> q) & 1u) && deg) {
      const u64 slot = (base >> BFS_VSHIFT) + (ex >> 40);
      out_row[slot] = ro[q];
      out_off[slot] = (u32)((base & BFS_EMASK) + (ex & DEGMASK));
      out_du[slot] = (u32)(i0 + q);
      ex += CNT1 | (u64)deg;
    }
  }
}

// dist_out[old_of_new[v]] = dist_layout[v]
__global__ __launch_bounds__(BLOCK) void k_sssp_unpermute(const u32* __restrict__ dist_layout, const int* __restrict__ old_of_new,
                                                         u32* __restrict__ dist_out, long long n) {
  for (long long v = (long long)blockIdx.x * BLOCK + threadIdx.x; v < n; v += (long long)gridDim.x * BLOCK)
    dist_out[old_of_new[v]] = dist_layout[v];
}

struct sssp_fused_state_t {
  mem_t<unsigned char> mark;
  mem_t<u32> dist_layout;            // only with a layout: distances in layout order
  mem_t<u32> q_row[2], q_off[2], q_du[2];
  mem_t<u32> frontier_bits;          // the frontier as a bitmap (the sweep; allocated on demand)
  unsigned dense_div = 4;            // an iteration whose frontier holds >= m / dense_div edges sweeps the unit blocks (sssp_dense_*; 0: never)
  mem_t<bfs_ctrl_t> ctrl;
  bfs_ctrl_t* host_ctrl = nullptr;
  int n = 0;
  int iters_hint = 12;
  float delta = 0.f;                 // near / far bucket width (0: off) (mgx_sssp_run_delta)
  // per-launch timing of k_sssp_relax (measurement runs: mgx_sssp_set_kernel_timing; an event costs ~6 us of stream gap)
  bool time_kernels = false;
  std::vector<hipEvent_t> ev;        // pairs around the relax launches of a batch, made on demand
  double relax_ms = 0.0;             // of the last run
  long long relax_launches = 0;
  sssp_fused_state_t(int num_nodes, standard_context_t& ctx) : n(num_nodes) {
    mark = mem_t<unsigned char>((size_t)num_nodes + 64, ctx);
    dist_layout = mem_t<u32>((size_t)num_nodes + 4, ctx);
    for (int i = 0; i < 2; ++i) {
      q_row[i] = mem_t<u32>((size_t)num_nodes + 1, ctx);
      q_off[i] = mem_t<u32>((size_t)num_nodes + 1, ctx);
      q_du[i] = mem_t<u32>((size_t)num_nodes + 1, ctx);
    }
    ctrl = mem_t<bfs_ctrl_t>(1, ctx);
    MGX_HIP(hipHostMalloc((void**)&host_ctrl, sizeof(bfs_ctrl_t), hipHostMallocDefault));
  }
  sssp_fused_state_t(const sssp_fused_state_t&) = delete;
  sssp_fused_state_t& operator=(const sssp_fused_state_t&) = delete;
  ~sssp_fused_state_t() {
    if (host_ctrl) (void)hipHostFree(host_ctrl);
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
  }
};

// Whole run from `src`; d_dist (n floats) holds the distances afterwards (+inf: unreachable is reported as the
// reference does, see the caller).  Returns with the stream synchronised and host_ctrl filled.
inline void sssp_fused_run(sssp_fused_state_t& st, const int* row_offsets, const int* col_indices, const float* weights,
                           float* d_dist, int src, standard_context_t& ctx, const sssp_layout_t* layout = nullptr) {
  hipStream_t s = ctx.stream();
  sssp_args_t a;
  a.row_offsets = (const u32*)(layout ? layout->row_offsets : row_offsets);
  a.col_indices = layout ? layout->col_indices : col_indices;
  a.weights = layout ? layout->weights : weights;
  a.dist = layout ? st.dist_layout.data() : (u32*)d_dist;
  a.mark = st.mark.data();
  for (int i = 0; i < 2; ++i) { a.q_row[i] = st.q_row[i].data(); a.q_off[i] = st.q_off[i].data(); a.q_du[i] = st.q_du[i].data(); }
  a.ctrl = st.ctrl.data();
  a.n = st.n;
  a.delta = st.delta;
  const char* const bl = mgx::env("MGX_SSSP_BUILD_LIST");            // (=1: the list-based queue build, k_sssp_build)
  const bool build2 = !(bl && atoi(bl) != 0) && ((uintptr_t)a.row_offsets % 16 == 0);
  a.m_edges = 0ull;
  a.frontier_bits = nullptr;
  // heavy iterations as a sweep over the layout's unit blocks and degree classes (sssp_dense_*): needs the frontier as a
  // bitmap (k_sssp_build2 writes it) and the plain loop (near / far buckets park vertices outside the queue)
  unsigned ddiv = st.dense_div;
  if (const char* e = mgx::env("MGX_SSSP_DENSE")) ddiv = (unsigned)atoi(e);
  const bool dense = layout && layout->ub_w && (layout->ub_col || layout->ub_col24) && layout->ub_cnt && layout->ub_owner && layout->ub_units_pad >= 16 &&
                     layout->vs_v[3] >= layout->vs_v[0] && layout->m_edges > 0 && build2 && ddiv > 0 && a.delta == 0.f;
  if (dense && !st.frontier_bits.size()) st.frontier_bits = mem_t<u32>(((size_t)st.n + 31) / 32 + 4, ctx);
  a.ub_col = dense ? layout->ub_col : nullptr;
  a.ub_w = dense ? layout->ub_w : nullptr;
  a.ub_cnt = dense ? layout->ub_cnt : nullptr;
  a.ub_owner = dense ? layout->ub_owner : nullptr;
  {
    // 24-bit ids whenever the layout has them (it has then dropped the 32-bit ones); half weights with them when every weight is
    // exact that way.  (MGX_SSSP_PACK, the round-4 A/B switch, is gone: 1.864 -> 1.790 ms per source, HISTORY.md 3.2b.)
    const bool pack = dense && layout->ub_col24;
    a.ub_col24 = pack ? layout->ub_col24 : nullptr;
    a.ub_w16 = (pack && layout->ub_w16) ? layout->ub_w16 : nullptr;
  }
  a.ub_units_pad = dense ? layout->ub_units_pad : 0u;
  for (int i = 0; i < 4; ++i) a.vs_v[i] = dense ? layout->vs_v[i] : 0u;
  a.dense_div = dense ? ddiv : 0u;
  if (dense) { a.m_edges = (unsigned long long)layout->m_edges; a.frontier_bits = st.frontier_bits.data(); }
  const char* const hme = mgx::env("MGX_SSSP_HOT_MIN_EDGES");        // (tests force the LDS bounds on small graphs)
  a.hot_min_edges = hme ? (u32)atoll(hme) : SSSP_HOT_MIN_EDGES;
  hipLaunchKernelGGL(k_sssp_init, dim3(grid_for(((long long)st.n + 3) / 4, BLOCK, ctx.num_cus * 8)), dim3(BLOCK), 0, s, a, src,
                     layout ? layout->new_of_old : (const int*)nullptr);
  static unsigned char attr_seen[64] = {};
  if (device_once_t once{attr_seen}) {
    MGX_HIP(hipFuncSetAttribute((const void*)k_sssp_relax<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, SSSP_HOTN * 2));
    MGX_HIP(hipFuncSetAttribute((const void*)(k_sssp_relax_dense<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  }
  int it = 0;
  st.relax_ms = 0.0;
  st.relax_launches = 0;
  for (int batch = 0;; ++batch) {
    const int nit = batch == 0 ? st.iters_hint : 2;
    if (st.time_kernels)
      while ((int)st.ev.size() < 2 * nit) { hipEvent_t e; MGX_HIP(hipEventCreate(&e)); st.ev.push_back(e); }
    for (int i = 0; i < nit; ++i, ++it) {
      if (st.time_kernels) MGX_HIP(hipEventRecord(st.ev[2 * i], s));
      hipLaunchKernelGGL(k_sssp_relax<1024>, dim3(ctx.num_cus * 2), dim3(1024), SSSP_HOTN * 2, s, a, it);
      if (dense) hipLaunchKernelGGL((k_sssp_relax_dense<1024>), dim3(ctx.num_cus), dim3(1024), SSSP_HOTN_DENSE * 2, s, a, it);
      if (st.time_kernels) MGX_HIP(hipEventRecord(st.ev[2 * i + 1], s));
      if (build2) hipLaunchKernelGGL(k_sssp_build2<512>, dim3(bfs_build_grid(st.n, 512)), dim3(512), 0, s, a, it);
      else hipLaunchKernelGGL(k_sssp_build<512>, dim3(bfs_build_grid(st.n, 512)), dim3(512), 0, s, a, it);
    }
    MGX_CHECK_LAUNCH("fused SSSP: kernel launch");
    MGX_HIP(hipMemcpyAsync(st.host_ctrl, st.ctrl.data(), offsetof(bfs_ctrl_t, trace) + 64 * sizeof(u64), hipMemcpyDeviceToHost, s));
    MGX_HIP(hipStreamSynchronize(s));
    for (int i = 0; st.time_kernels && i < nit; ++i) {
      float ms = 0.f;
      MGX_HIP(hipEventElapsedTime(&ms, st.ev[2 * i], st.ev[2 * i + 1]));
      st.relax_ms += ms;
      st.relax_launches += 1;
    }
    if (st.host_ctrl->done) break;
    // every iteration launched so far has run: an empty next frontier ends the loop here, without the launch that
    // would find that out
    if ((st.host_ctrl->cursor[it % 3] >> BFS_VSHIFT) == 0 && st.host_ctrl->sssp_far_cnt[it & 1] == 0) {
      st.host_ctrl->done = 1;
      st.host_ctrl->levels = it;
      break;
    }
  }
  st.iters_hint = st.host_ctrl->levels > 0 ? st.host_ctrl->levels : 1;
  if (layout) {
    hipLaunchKernelGGL(k_sssp_unpermute, dim3(grid_for(st.n, BLOCK, ctx.num_cus * 8)), dim3(BLOCK), 0, s, st.dist_layout.data(),
                       layout->old_of_new, (u32*)d_dist, (long long)st.n);
    MGX_CHECK_LAUNCH("fused SSSP: unpermute launch");
    MGX_HIP(hipStreamSynchronize(s));
  }
}

}  // namespace mgx
