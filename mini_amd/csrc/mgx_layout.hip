// mgx_layout.hip -- hub-first layout of a device CSR: vertices renumbered by descending degree (stable:
// ties keep their original order), rows rebuilt under the new numbering with their neighbour lists sorted.
// Graph construction, not the hot path: one-time setup per graph, so it leans on rocPRIM's device-wide sort / scan
// instead of kernels of its own.  Same result as mini_amd.rmat.degree_order (torch device ops), which the tests
// compare it with.  A translation unit of its own: rocPRIM's headers are heavy, and nothing else needs them.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_reduce.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_segmented_radix_sort.hpp>

#include <cstdint>
#include <cstdio>

namespace {

struct tmp_t {
  void* p = nullptr;
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  ~tmp_t() { if (p) (void)hipFree(p); }
  template <typename T> T* as() const { return (T*)p; }
};

#define LAY_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return (int)e_; } while (0)

__global__ void k_degrees(const int* __restrict__ ro, int n, int* __restrict__ deg, int* __restrict__ ids) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { deg[i] = ro[i + 1] - ro[i]; ids[i] = (int)i; }
}

__global__ void k_invert(const int* __restrict__ old_of_new, int n, int* __restrict__ new_of_old) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) new_of_old[old_of_new[i]] = (int)i;
}

// entry e of the NEW CSR: its row by a search in the new offsets (setup code: 22 probes per entry are fine), then
// the matching entry of the old row, neighbour renumbered
__global__ void k_fill(const int* __restrict__ lro, const int* __restrict__ old_of_new, const int* __restrict__ new_of_old,
                       const int* __restrict__ ro, const int* __restrict__ ci, const float* __restrict__ w, int n,
                       long long m, int* __restrict__ lci, float* __restrict__ lw) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += (long long)gridDim.x * blockDim.x) {
    int lo = 0, hi = n;                       // last row with lro[row] <= e
    while (hi - lo > 1) {
      const int mid = lo + (hi - lo) / 2;
      if ((long long)lro[mid] <= e) lo = mid; else hi = mid;
    }
    const long long src = (long long)ro[old_of_new[lo]] + (e - lro[lo]);
    lci[e] = new_of_old[ci[src]];
    if (w) lw[e] = w[src];
  }
}

}  // namespace

// All pointers are device pointers; lro has n + 1 entries, lci / lw m, the two maps n.  w / lw may be NULL.
// Runs on `stream` and returns with it synchronised.  Returns 0 or the hipError_t that stopped it.
extern "C" int mgx_layout_build_device(const int* ro, const int* ci, const float* w, int n, long long m, int* lro, int* lci,
                                       float* lw, int* new_of_old, int* old_of_new, hipStream_t stream) {
  if (n <= 0) return 0;
  const int threads = 256;
  const unsigned nblocks = (unsigned)(((long long)n + threads - 1) / threads);
  tmp_t deg, ids, deg_sorted, scratch;
  LAY_TRY(deg.alloc((size_t)n * 4)); LAY_TRY(ids.alloc((size_t)n * 4)); LAY_TRY(deg_sorted.alloc((size_t)n * 4));
  hipLaunchKernelGGL(k_degrees, dim3(nblocks), dim3(threads), 0, stream, ro, n, deg.as<int>(), ids.as<int>());

  // vertices by descending degree (radix sort: stable)
  size_t bytes = 0;
  LAY_TRY(rocprim::radix_sort_pairs_desc(nullptr, bytes, deg.as<int>(), deg_sorted.as<int>(), ids.as<int>(), old_of_new,
                                         (size_t)n, 0, 32, stream));
  LAY_TRY(scratch.alloc(bytes));
  LAY_TRY(rocprim::radix_sort_pairs_desc(scratch.p, bytes, deg.as<int>(), deg_sorted.as<int>(), ids.as<int>(), old_of_new,
                                         (size_t)n, 0, 32, stream));
  hipLaunchKernelGGL(k_invert, dim3(nblocks), dim3(threads), 0, stream, old_of_new, n, new_of_old);

  // new row offsets: exclusive scan of the sorted degrees, lro[n] = m
  {
    size_t sb = 0;
    LAY_TRY(rocprim::exclusive_scan(nullptr, sb, deg_sorted.as<int>(), lro, 0, (size_t)n, rocprim::plus<int>(), stream));
    tmp_t st; LAY_TRY(st.alloc(sb));
    LAY_TRY(rocprim::exclusive_scan(st.p, sb, deg_sorted.as<int>(), lro, 0, (size_t)n, rocprim::plus<int>(), stream));
    const int mm = (int)m;
    LAY_TRY(hipMemcpyAsync(lro + n, &mm, sizeof(int), hipMemcpyHostToDevice, stream));
    LAY_TRY(hipStreamSynchronize(stream));       // (mm lives on this frame; st is freed here)
  }
  if (m <= 0) return 0;

  // rows under the new numbering, then every row sorted by neighbour id
  tmp_t ci_tmp, w_tmp;
  LAY_TRY(ci_tmp.alloc((size_t)m * 4));
  if (w) LAY_TRY(w_tmp.alloc((size_t)m * 4));
  hipLaunchKernelGGL(k_fill, dim3(8192), dim3(threads), 0, stream, lro, old_of_new, new_of_old, ro, ci, w, n, m,
                     ci_tmp.as<int>(), w ? w_tmp.as<float>() : (float*)nullptr);
  int bits = 1;
  while (bits < 32 && (1ll << bits) < (long long)n) ++bits;
  size_t sb = 0;
  tmp_t st;
  if (w) {
    LAY_TRY(rocprim::segmented_radix_sort_pairs(nullptr, sb, ci_tmp.as<int>(), lci, w_tmp.as<float>(), lw, (size_t)m,
                                                (unsigned)n, lro, lro + 1, 0, bits, stream));
    LAY_TRY(st.alloc(sb));
    LAY_TRY(rocprim::segmented_radix_sort_pairs(st.p, sb, ci_tmp.as<int>(), lci, w_tmp.as<float>(), lw, (size_t)m,
                                                (unsigned)n, lro, lro + 1, 0, bits, stream));
  } else {
    LAY_TRY(rocprim::segmented_radix_sort_keys(nullptr, sb, ci_tmp.as<int>(), lci, (size_t)m, (unsigned)n, lro, lro + 1, 0,
                                               bits, stream));
    LAY_TRY(st.alloc(sb));
    LAY_TRY(rocprim::segmented_radix_sort_keys(st.p, sb, ci_tmp.as<int>(), lci, (size_t)m, (unsigned)n, lro, lro + 1, 0,
                                               bits, stream));
  }
  LAY_TRY(hipStreamSynchronize(stream));
  return 0;
}

// ---- unit-blocked copy of a degree class ("unit blocks") -------------------------------------------------
// The rows of a CSR whose degree lies in [min_deg, max_deg) copied so that every row starts on a multiple of
// U = 1 << ushift entries and is padded to a multiple of U (padding is -1: the traversal kernels' LDS bitmap has a word of
// ones in front, so a -1 reads as visited; a reduction takes it for the identity), plus owner[u] = the row unit u belongs to.  A unit is U consecutive entries of ONE
// row: a kernel can stream the copy in fixed-size pieces (bfs_fused_dense.hpp) and decide per unit whether its row is in
// the frontier, with no row walk and no search.  The number of units is padded to a multiple of 16 (owner = n for the
// padding: a vertex that is never in a frontier) and the copy ends with four entries of -1 (where lanes of inactive
// units read from).
namespace {

// entries of the sorted row [r0, r1) that are below `limit` (0: all of them)
__device__ __forceinline__ int entries_below(const int* __restrict__ ci, int r0, int r1, unsigned limit) {
  if (limit == 0u) return r1 - r0;
  int lo = r0, hi = r1;
  while (lo < hi) {
    const int mid = lo + ((hi - lo) >> 1);
    if ((unsigned)ci[mid] < limit) lo = mid + 1; else hi = mid;
  }
  return lo - r0;
}

__global__ void k_unit_counts(const int* __restrict__ ro, const int* __restrict__ ci, int n, int min_deg, int max_deg, int ushift,
                              unsigned hot_limit, int* __restrict__ cnt) {
  const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n) return;
  const int deg = ro[v + 1] - ro[v];
  const int U = 1 << ushift;
  const int kept = (deg >= min_deg && deg < max_deg) ? entries_below(ci, ro[v], ro[v + 1], hot_limit) : 0;
  cnt[v] = (kept + U - 1) >> ushift;
}

// one wave per row of the class: owners and padded entries
__global__ void k_unit_fill(const int* __restrict__ ro, const int* __restrict__ ci, int n, const int* __restrict__ uoff,
                            int ushift, unsigned hot_limit, int* __restrict__ owner, int* __restrict__ ucol, unsigned char* __restrict__ ucnt) {
  const int lane = threadIdx.x & 63;
  const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long v = wave0; v < n; v += nwaves) {
    const int u0 = uoff[v], u1 = uoff[v + 1];
    if (u1 == u0) continue;
    const int r0 = ro[v], deg = entries_below(ci, r0, ro[v + 1], hot_limit);      // (hot_limit: the row's entries below it only -- a prefix: rows are sorted)
    for (int u = u0 + lane; u < u1; u += 64) {
      owner[u] = (int)v;
      // real entries of the unit (the rest is padding, -1): the fused SSSP's sweep masks by it (mgx/sssp_fused.hpp)
      const long long left = (long long)deg - ((long long)(u - u0) << ushift);
      ucnt[u] = (unsigned char)(left < (1 << ushift) ? left : (1 << ushift));
    }
    const long long e0 = (long long)u0 << ushift, e1 = (long long)u1 << ushift;
    for (long long e = e0 + lane; e < e1; e += 64) {
      const long long k = e - e0;
      ucol[e] = k < deg ? ci[r0 + k] : -1;        // -1: "visited" to a traversal (the sentinel word in front of the LDS bitmap), the identity to a reduction
    }
  }
}

__global__ void k_unit_tail(int n, int units, int units_pad, int ushift, int* __restrict__ owner, int* __restrict__ ucol,
                            unsigned char* __restrict__ ucnt) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long tail_entries = ((long long)(units_pad - units) << ushift) + 4;
  if (i < units_pad - units) { owner[units + i] = n; ucnt[units + i] = 0; }
  if (i < tail_entries) ucol[((long long)units << ushift) + i] = -1;
}

}  // namespace

// Allocates *owner (units_pad ints) and *ucol ((units_pad << ushift) + 4 ints) with hipMalloc; the caller owns them.
// *units = real units, *units_pad = padded to a multiple of 16.  *ucnt (units_pad bytes): real entries of every unit (ushift
// <= 7); *ufirst (n + 1 ints): the units of row v are [ufirst[v], ufirst[v + 1]).  Returns 0 or the hipError_t that stopped it.
// hot_limit != 0 (rows SORTED by neighbour id): only the entries below it are copied -- a rank of the partitioned traversal whose
// other entries live in the cold-edge lists (bfs_fused_cold.hpp): its unit-block body would read them only to skip them.
extern "C" int mgx_units_build_device(const int* ro, const int* ci, int n, int min_deg, int max_deg, int ushift, unsigned hot_limit,
                                      int** owner, int** ucol, unsigned char** ucnt, int** ufirst, long long* units, long long* units_pad,
                                      hipStream_t stream) {
  *owner = nullptr; *ucol = nullptr; *ucnt = nullptr; *ufirst = nullptr; *units = 0; *units_pad = 0;
  if (n <= 0) return 0;
  const int threads = 256;
  const unsigned nblocks = (unsigned)(((long long)n + threads - 1) / threads);
  tmp_t cnt, uoff, st;
  LAY_TRY(cnt.alloc((size_t)n * 4)); LAY_TRY(uoff.alloc(((size_t)n + 1) * 4));
  hipLaunchKernelGGL(k_unit_counts, dim3(nblocks), dim3(threads), 0, stream, ro, ci, n, min_deg, max_deg, ushift, hot_limit, cnt.as<int>());
  size_t sb = 0;
  LAY_TRY(rocprim::exclusive_scan(nullptr, sb, cnt.as<int>(), uoff.as<int>(), 0, (size_t)n, rocprim::plus<int>(), stream));
  LAY_TRY(st.alloc(sb));
  LAY_TRY(rocprim::exclusive_scan(st.p, sb, cnt.as<int>(), uoff.as<int>(), 0, (size_t)n, rocprim::plus<int>(), stream));
  int last_off = 0, last_cnt = 0;
  LAY_TRY(hipMemcpyAsync(&last_off, uoff.as<int>() + (n - 1), 4, hipMemcpyDeviceToHost, stream));
  LAY_TRY(hipMemcpyAsync(&last_cnt, cnt.as<int>() + (n - 1), 4, hipMemcpyDeviceToHost, stream));
  LAY_TRY(hipStreamSynchronize(stream));
  const long long U = (long long)last_off + last_cnt;
  if (U <= 0) return 0;
  if ((U << ushift) > 2000000000LL) return 0;              // 32-bit entry offsets in the kernels: no unit blocks then
  const int tot = (int)U;
  LAY_TRY(hipMemcpyAsync(uoff.as<int>() + n, &tot, 4, hipMemcpyHostToDevice, stream));
  const long long Up = (U + 15) / 16 * 16;
  LAY_TRY(hipMalloc((void**)owner, (size_t)Up * 4));
  hipError_t e = hipMalloc((void**)ucol, (((size_t)Up << ushift) + 4) * 4);
  if (e != hipSuccess) { (void)hipFree(*owner); *owner = nullptr; return (int)e; }
  e = hipMalloc((void**)ucnt, (size_t)Up + 16);
  if (e == hipSuccess) e = hipMalloc((void**)ufirst, ((size_t)n + 1) * 4);
  if (e != hipSuccess) {
    (void)hipFree(*owner); (void)hipFree(*ucol); if (*ucnt) (void)hipFree(*ucnt);
    *owner = nullptr; *ucol = nullptr; *ucnt = nullptr; *ufirst = nullptr;
    return (int)e;
  }
  hipLaunchKernelGGL(k_unit_fill, dim3(4096), dim3(256), 0, stream, ro, ci, n, uoff.as<int>(), ushift, hot_limit, *owner, *ucol, *ucnt);
  const long long tail = ((Up - U) << ushift) + 4;
  hipLaunchKernelGGL(k_unit_tail, dim3((unsigned)((tail + 255) / 256)), dim3(256), 0, stream, n, (int)U, (int)Up, ushift, *owner, *ucol, *ucnt);
  (void)hipMemcpyAsync(*ufirst, uoff.as<int>(), ((size_t)n + 1) * 4, hipMemcpyDeviceToDevice, stream);
  e = hipStreamSynchronize(stream);                       // (tot lives on this frame)
  if (e != hipSuccess) {
    (void)hipFree(*owner); (void)hipFree(*ucol); (void)hipFree(*ucnt); (void)hipFree(*ufirst);
    *owner = nullptr; *ucol = nullptr; *ucnt = nullptr; *ufirst = nullptr;
    return (int)e;
  }
  *units = U; *units_pad = Up;
  return 0;
}

// ---- cold-edge lists of the long rows (mgx/bfs_fused_cold.hpp) --------------------------------------------------------
// The unit-block body keeps the first `hot_n` vertices of the visited bitmap in LDS; an entry that points behind them is
// "cold": it cannot be tested there, and marking it costs a scattered one-byte store -- which, for a 4 MB mark array under
// a stream, costs the fabric as much as a whole cache line (tools/microbench4.hip).  Here the cold entries of the rows
// [0, rows) are pulled out once, as (owner, dst) pairs grouped by the SLICE of the id range dst lies in (slice k =
// [hot_n + k * slice_n, hot_n + (k + 1) * slice_n)): a workgroup that takes pairs of ONE slice can keep that slice of
// the bitmap in LDS and needs no marks at all.  Inside a slice the pairs are ordered by owner, then dst (the rows are
// sorted): neighbouring lanes ask for neighbouring frontier bits.
namespace {

// cnt[k * rows + r] = entries of row r in slice k (0 for rows of fewer than min_deg entries)
__global__ void k_cold_counts(const int* __restrict__ ro, const int* __restrict__ ci, int row0, int rows, int min_deg, unsigned hot_n,
                              unsigned slice_n, int slices, int* __restrict__ cnt) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= rows) return;
  const int r0 = ro[row0 + r], r1 = ro[row0 + r + 1];
  const bool is_long = r1 - r0 >= min_deg;
  int prev = r1;
  if (is_long) {                                   // first entry >= hot_n
    int lo = r0, hi = r1;
    while (lo < hi) { const int mid = lo + (hi - lo) / 2; if ((unsigned)ci[mid] < hot_n) lo = mid + 1; else hi = mid; }
    prev = lo;
  }
  for (int k = 0; k < slices; ++k) {
    int next = r1;
    if (is_long && k + 1 < slices) {
      const unsigned long long bound = (unsigned long long)hot_n + (unsigned long long)(k + 1) * slice_n;
      int lo = prev, hi = r1;
      while (lo < hi) { const int mid = lo + (hi - lo) / 2; if ((unsigned long long)(unsigned)ci[mid] < bound) lo = mid + 1; else hi = mid; }
      next = lo;
    }
    cnt[(long long)k * rows + r] = is_long ? next - prev : 0;
    prev = next;
  }
}

// one wave per row: its cold entries to their places (off[k * rows + r] = first pair of (slice k, row r))
__global__ void k_cold_fill(const int* __restrict__ ro, const int* __restrict__ ci, int row0, int rows, const int* __restrict__ cnt,
                            const int* __restrict__ off, int slices, int* __restrict__ owner, int* __restrict__ dst) {
  const int lane = threadIdx.x & 63;
  const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long r = wave0; r < rows; r += nwaves) {
    int total = 0;
    for (int k = 0; k < slices; ++k) total += cnt[(long long)k * rows + r];
    if (total == 0) continue;
    int src = ro[row0 + r + 1] - total;            // the cold entries are the row's tail
    for (int k = 0; k < slices; ++k) {
      const int c = cnt[(long long)k * rows + r], o = off[(long long)k * rows + r];
      for (int i = lane; i < c; i += 64) { owner[o + i] = row0 + (int)r; dst[o + i] = ci[src + i]; }
      src += c;
    }
  }
}

__global__ void k_cold_pad(int pairs, int pad, int n, int* __restrict__ owner, int* __restrict__ dst) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < pad) { owner[pairs + i] = n; dst[pairs + i] = -1; }
}

}  // namespace

// The rows [row0, row0 + rows) of at least min_deg entries.  slice_off: slices + 1 ints (host).  Allocates *owner / *dst
// (pairs + 256 ints each: the padding holds owner = n, dst = -1) with hipMalloc; nothing is allocated when there are no pairs.
extern "C" int mgx_cold_build_device(const int* ro, const int* ci, int n, int row0, int rows, int min_deg, unsigned hot_n, unsigned slice_n,
                                     int slices, int** owner, int** dst, long long* pairs, int* slice_off, hipStream_t stream) {
  *owner = nullptr; *dst = nullptr; *pairs = 0;
  for (int k = 0; k <= slices; ++k) slice_off[k] = 0;
  if (rows <= 0 || slices <= 0) return 0;
  const size_t cells = (size_t)rows * (size_t)slices;
  tmp_t cnt, off, st;
  LAY_TRY(cnt.alloc(cells * 4)); LAY_TRY(off.alloc(cells * 4));
  hipLaunchKernelGGL(k_cold_counts, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, ro, ci, row0, rows, min_deg, hot_n,
                     slice_n, slices, cnt.as<int>());
  size_t sb = 0;
  LAY_TRY(rocprim::exclusive_scan(nullptr, sb, cnt.as<int>(), off.as<int>(), 0, cells, rocprim::plus<int>(), stream));
  LAY_TRY(st.alloc(sb));
  LAY_TRY(rocprim::exclusive_scan(st.p, sb, cnt.as<int>(), off.as<int>(), 0, cells, rocprim::plus<int>(), stream));
  int last_off = 0, last_cnt = 0;
  LAY_TRY(hipMemcpyAsync(&last_off, off.as<int>() + (cells - 1), 4, hipMemcpyDeviceToHost, stream));
  LAY_TRY(hipMemcpyAsync(&last_cnt, cnt.as<int>() + (cells - 1), 4, hipMemcpyDeviceToHost, stream));
  for (int k = 0; k < slices; ++k)
    LAY_TRY(hipMemcpyAsync(slice_off + k, off.as<int>() + (size_t)k * rows, 4, hipMemcpyDeviceToHost, stream));
  LAY_TRY(hipStreamSynchronize(stream));
  const long long E = (long long)last_off + last_cnt;
  slice_off[slices] = (int)E;
  if (E <= 0) return 0;
  LAY_TRY(hipMalloc((void**)owner, ((size_t)E + 256) * 4));
  hipError_t e = hipMalloc((void**)dst, ((size_t)E + 256) * 4);
  if (e != hipSuccess) { (void)hipFree(*owner); *owner = nullptr; return (int)e; }
  hipLaunchKernelGGL(k_cold_fill, dim3(2048), dim3(256), 0, stream, ro, ci, row0, rows, cnt.as<int>(), off.as<int>(), slices, *owner, *dst);
  hipLaunchKernelGGL(k_cold_pad, dim3(1), dim3(256), 0, stream, (int)E, 256, n, *owner, *dst);
  e = hipStreamSynchronize(stream);
  if (e != hipSuccess) { (void)hipFree(*owner); (void)hipFree(*dst); *owner = nullptr; *dst = nullptr; return (int)e; }
  *pairs = E;
  return 0;
}

// ---- the pair lists, four bytes per pair ------------------------------------------------------------------------------
// A level that reads the unit blocks streams ALL pairs of every slice (bfs_fused_cold.hpp), eight bytes each: on a rank of
// RMAT-26 / 8 that is 712 MB per dense level -- 168 us, as much as its whole unit-block stream.  Inside a slice the pairs are
// ordered by owner, so 64 consecutive pairs (what a wave loads at once) span few owners: the copy made here keeps, per chunk of
// 64 pairs of a slice, the owner of its first pair (cbase) and per pair
//     bits 0..19  dst - first vertex of the slice      (a slice is 652 288 vertices)
//     bits 20..31 (owner - cbase of the chunk) / ranks (owners of a rank's list are global ids of ITS rows: multiples apart)
// A slice with a chunk that spans 4 096 owners or more keeps the 8-byte pairs (its bit in *mask stays 0).
namespace {
struct cold_pack_t {
  unsigned off[65];      // pairs of slice q: [off[q], off[q + 1])
  unsigned lo[64];       // its first vertex
  unsigned cb[65];       // its chunks' owners: cbase[cb[q] ..)
  int used;
  int ranks;
};
__global__ void k_cold_pack(const int* __restrict__ owner, const int* __restrict__ dst, cold_pack_t P, unsigned* __restrict__ pk,
                            unsigned* __restrict__ cbase, unsigned* __restrict__ bad) {
  const unsigned total = P.off[P.used];
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total + 256u; i += gridDim.x * blockDim.x) {
    if (i >= total) { pk[i] = 0u; continue; }
    int q = 0;
    while (q + 1 < P.used && i >= P.off[q + 1]) ++q;
    const unsigned p0 = P.off[q], c = (i - p0) >> 6;
    const unsigned base = (unsigned)owner[p0 + (c << 6)];
    const unsigned own = (unsigned)owner[i];
    const unsigned delta = (own - base) / (unsigned)P.ranks;
    const unsigned rel = (unsigned)dst[i] - P.lo[q];
    if (own < base || delta >= 4096u || rel >= (1u << 20) || (own - base) % (unsigned)P.ranks) atomicOr(&bad[q >> 5], 1u << (q & 31));
    pk[i] = (rel & 0xFFFFFu) | (delta << 20);
    if (((i - p0) & 63u) == 0u) cbase[P.cb[q] + c] = base;
  }
}
}  // namespace

// owner / dst: the lists of mgx_cold_build_device AFTER the caller has put the owners into the id space the kernels look them
// up in; used <= 64 slices, off (used + 1) / lo (used) as the kernels get them.  Allocates *pk (pairs + 256 words) and *cbase;
// cb_off (used + 1, host): where a slice's chunk owners start; *mask: bit q set = slice q is packed.
extern "C" int mgx_cold_pack_device(const int* owner, const int* dst, int used, const unsigned* off, const unsigned* lo, int ranks,
                                    unsigned** pk, unsigned** cbase, unsigned* cb_off, unsigned long long* mask, hipStream_t stream) {
  *pk = nullptr; *cbase = nullptr; *mask = 0ull;
  if (used <= 0 || used > 64 || ranks <= 0) return 0;
  cold_pack_t P;
  P.used = used; P.ranks = ranks;
  unsigned acc = 0;
  for (int q = 0; q < used; ++q) { P.off[q] = off[q]; P.lo[q] = lo[q]; P.cb[q] = acc; cb_off[q] = acc; acc += (off[q + 1] - off[q] + 63u) / 64u; }
  P.off[used] = off[used]; P.cb[used] = acc; cb_off[used] = acc;
  for (int q = used + 1; q <= 64; ++q) { P.off[q] = off[used]; P.cb[q] = acc; }
  for (int q = used; q < 64; ++q) P.lo[q] = 0;
  const unsigned total = off[used];
  if (total == 0u) return 0;
  tmp_t bad;
  LAY_TRY(bad.alloc(8));
  LAY_TRY(hipMemsetAsync(bad.p, 0, 8, stream));
  LAY_TRY(hipMalloc((void**)pk, ((size_t)total + 256) * 4));
  hipError_t e = hipMalloc((void**)cbase, ((size_t)acc + 64) * 4);
  if (e != hipSuccess) { (void)hipFree(*pk); *pk = nullptr; return (int)e; }
  (void)hipMemsetAsync(*cbase, 0, ((size_t)acc + 64) * 4, stream);
  hipLaunchKernelGGL(k_cold_pack, dim3(4096), dim3(256), 0, stream, owner, dst, P, *pk, *cbase, bad.as<unsigned>());
  unsigned hb[2] = {0, 0};
  e = hipMemcpyAsync(hb, bad.p, 8, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) { (void)hipFree(*pk); (void)hipFree(*cbase); *pk = nullptr; *cbase = nullptr; return (int)e; }
  const unsigned long long badmask = (unsigned long long)hb[0] | ((unsigned long long)hb[1] << 32);
  const unsigned long long all = used == 64 ? ~0ull : ((1ull << used) - 1ull);
  *mask = all & ~badmask;
  return 0;
}

// ---- genuine CSC (transpose) of a device CSR ------------------------------------------------------------------------
// The reference's loader always ends up with csc == csr (its transposed copy goes into a shadowed local), which
// is only right for symmetric inputs.  Bottom-up BFS levels on a DIRECTED graph need the in-edges: col_offsets[v] ..
// col_offsets[v + 1] index row_indices (the sources of v's in-edges, ascending) and row_values (their weights).
// One stable radix sort of the edge numbers by destination: sources come out ascending inside every column because the
// CSR lists its edges by ascending row.
namespace {

__global__ void k_edge_rows(const int* __restrict__ ro, int n, long long m, int* __restrict__ row_of_edge, int* __restrict__ edge_id) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += (long long)gridDim.x * blockDim.x) {
    int lo = 0, hi = n;                       // last row with ro[row] <= e
    while (hi - lo > 1) {
      const int mid = lo + (hi - lo) / 2;
      if ((long long)ro[mid] <= e) lo = mid; else hi = mid;
    }
    row_of_edge[e] = lo;
    edge_id[e] = (int)e;
  }
}

__global__ void k_csc_gather(const int* __restrict__ sorted_edge, const int* __restrict__ row_of_edge, const float* __restrict__ w,
                             long long m, int* __restrict__ row_indices, float* __restrict__ row_values) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x) {
    const int e = sorted_edge[i];
    row_indices[i] = row_of_edge[e];
    if (row_values) row_values[i] = w ? w[e] : 1.0f;
  }
}

// col_offsets[v] = first position of the sorted destinations that is >= v
__global__ void k_csc_offsets(const int* __restrict__ sorted_dst, long long m, int n, int* __restrict__ col_offsets) {
  const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (v > n) return;
  long long lo = 0, hi = m;                   // first i with sorted_dst[i] >= v
  while (lo < hi) {
    const long long mid = lo + (hi - lo) / 2;
    if (sorted_dst[mid] < (int)v) lo = mid + 1; else hi = mid;
  }
  col_offsets[v] = (int)lo;
}

}  // namespace

// co: n + 1 ints, ri: m ints, rv: m floats or NULL (device, caller-allocated).  w may be NULL (unit weights).
extern "C" int mgx_csc_build_device(const int* ro, const int* ci, const float* w, int n, long long m, int* co, int* ri, float* rv,
                                    hipStream_t stream) {
  if (n < 0) return 0;
  if (m <= 0) { LAY_TRY(hipMemsetAsync(co, 0, ((size_t)n + 1) * 4, stream)); return (int)hipStreamSynchronize(stream); }
  tmp_t row_of_edge, edge_id, sorted_dst, sorted_edge, scratch;
  LAY_TRY(row_of_edge.alloc((size_t)m * 4)); LAY_TRY(edge_id.alloc((size_t)m * 4));
  LAY_TRY(sorted_dst.alloc((size_t)m * 4)); LAY_TRY(sorted_edge.alloc((size_t)m * 4));
  hipLaunchKernelGGL(k_edge_rows, dim3(8192), dim3(256), 0, stream, ro, n, m, row_of_edge.as<int>(), edge_id.as<int>());
  int bits = 1;
  while (bits < 32 && (1ll << bits) < (long long)n) ++bits;
  size_t bytes = 0;
  LAY_TRY(rocprim::radix_sort_pairs(nullptr, bytes, ci, sorted_dst.as<int>(), edge_id.as<int>(), sorted_edge.as<int>(), (size_t)m, 0,
                                    bits, stream));
  LAY_TRY(scratch.alloc(bytes));
  LAY_TRY(rocprim::radix_sort_pairs(scratch.p, bytes, ci, sorted_dst.as<int>(), edge_id.as<int>(), sorted_edge.as<int>(), (size_t)m,
                                    0, bits, stream));
  hipLaunchKernelGGL(k_csc_gather, dim3(8192), dim3(256), 0, stream, sorted_edge.as<int>(), row_of_edge.as<int>(), w, m, ri, rv);
  hipLaunchKernelGGL(k_csc_offsets, dim3((unsigned)(((long long)n + 1 + 255) / 256)), dim3(256), 0, stream, sorted_dst.as<int>(), m, n, co);
  LAY_TRY(hipStreamSynchronize(stream));
  return 0;
}

// ---- a rank's shard of the partitioned R-MAT graph (mgx/bfs_dist2.hpp; mgx_dbfs2_shard_*) ------------------------------
// What mini_amd.dist_bfs.rmat_cyclic_shard did with torch device ops, inside the library: the symmetrised R-MAT graph of
// (scale, edgefactor, seed) under the generation-2 layout -- vertices renumbered hub-first by GLOBAL degree (every rank
// derives the same permutation from the same counter-based pair stream: no communication), vertex v owned by rank
// v % ranks, local row v / ranks, neighbour ids global.
//   plan   one pass over the pair stream: global degrees (atomic adds), stable sort by descending degree -> old_of_new,
//          new_of_old; the rank's rows and entries are then known (n_local, m_local)
//   fill   a second pass: both directions of every pair whose (new) row this rank owns as keys (local row << 32 | neighbour)
//          into a buffer of exactly m_local keys, one 64-bit radix sort, split into row offsets and neighbour ids
// Every rank generates the WHOLE pair stream twice (RMAT-26: 2^30 pairs, ~40 hash rounds each: tens of milliseconds on one
// MI355X) instead of generating a share and shuffling 17 GB of pairs between the ranks.
#include "mgx/rmat.hpp"
namespace {

__global__ void k_shard_degrees(int scale, long long first, long long count, unsigned long long seed, int* __restrict__ deg) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
    unsigned s, d;
    mgx::rmat_pair(scale, seed, (unsigned long long)(first + i), 1, s, d);
    atomicAdd(deg + s, 1);                  // (the symmetrised CSR holds every pair in both directions)
    atomicAdd(deg + d, 1);
  }
}
__global__ void k_iota(int n, int* __restrict__ ids) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) ids[i] = (int)i;
}
// local row i of the rank = new vertex i * ranks + rank: its degree
__global__ void k_shard_local_degrees(const int* __restrict__ deg_sorted, int n, int ranks, int rank, int n_local, int* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_local) out[i] = deg_sorted[i * ranks + rank];
}
__global__ void k_shard_keys(int scale, long long first, long long count, unsigned long long seed, const int* __restrict__ new_of_old,
                             int ranks, int rank, unsigned long long* __restrict__ keys, unsigned long long* cursor,
                             unsigned long long cap) {
  const int lane = (int)(threadIdx.x & 63);
  for (long long i0 = (long long)blockIdx.x * blockDim.x; i0 < count; i0 += (long long)gridDim.x * blockDim.x) {   // (block-uniform trips)
    const long long i = i0 + threadIdx.x;
    unsigned long long k0 = 0, k1 = 0;
    int have = 0;
    if (i < count) {
      unsigned s, d;
      mgx::rmat_pair(scale, seed, (unsigned long long)(first + i), 1, s, d);
      const unsigned ns = (unsigned)new_of_old[s], nd = (unsigned)new_of_old[d];
      // pair (u, v): CSR row v holds u (graph.hxx F9 orientation), and the swapped copy: row u holds v
      if ((int)(nd % (unsigned)ranks) == rank) { k0 = ((unsigned long long)(nd / (unsigned)ranks) << 32) | ns; have = 1; }
      if ((int)(ns % (unsigned)ranks) == rank) { (have ? k1 : k0) = ((unsigned long long)(ns / (unsigned)ranks) << 32) | nd; have += 1; }
    }
    // wave-aggregated append
    int inc = have;
#pragma unroll
    for (int dlt = 1; dlt < 64; dlt <<= 1) { const int y = __shfl_up(inc, dlt, 64); if (lane >= dlt) inc += y; }
    const int tot = __shfl(inc, 63, 64);
    unsigned long long base = 0;
    if (lane == 63 && tot) base = atomicAdd(cursor, (unsigned long long)tot);
    base = __shfl(base, 63, 64);
    const unsigned long long at = base + (unsigned long long)(inc - have);
    if (have >= 1 && at < cap) keys[at] = k0;
    if (have == 2 && at + 1 < cap) keys[at + 1] = k1;
  }
}
__global__ void k_shard_split(const unsigned long long* __restrict__ keys, long long m, int* __restrict__ col) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (long long)gridDim.x * blockDim.x)
    col[i] = (int)(keys[i] & 0xFFFFFFFFull);
}

struct shard_plan_t {
  int scale = 0, edgefactor = 0, ranks = 1, rank = 0, n = 0, n_local = 0;
  unsigned long long seed = 0;
  long long m_local = 0;
  tmp_t deg_sorted, old_of_new, new_of_old, ro_local;
};

}  // namespace

// plan: *handle owns device arrays until mgx_shard_free_device; n_local / m_local tell the caller what to allocate
extern "C" int mgx_shard_plan_device(int scale, int edgefactor, unsigned long long seed, int ranks, int rank, void** handle, int* n_local,
                                     long long* m_local, hipStream_t stream) {
  *handle = nullptr;
  shard_plan_t* P = new shard_plan_t();
  P->scale = scale; P->edgefactor = edgefactor; P->seed = seed; P->ranks = ranks; P->rank = rank;
  const int n = 1 << scale;
  P->n = n;
  P->n_local = (n - rank + ranks - 1) / ranks;
  const long long total = (long long)edgefactor * n;
  tmp_t deg, ids;
  const int threads = 256;
  const unsigned nb = (unsigned)(((long long)n + threads - 1) / threads);
#define SHARD_TRY(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { delete P; return (int)e_; } } while (0)
  SHARD_TRY(deg.alloc((size_t)n * 4)); SHARD_TRY(ids.alloc((size_t)n * 4));
  SHARD_TRY(P->deg_sorted.alloc((size_t)n * 4)); SHARD_TRY(P->old_of_new.alloc((size_t)n * 4)); SHARD_TRY(P->new_of_old.alloc((size_t)n * 4));
  SHARD_TRY(hipMemsetAsync(deg.p, 0, (size_t)n * 4, stream));
  const long long chunk = 1ll << 28;
  for (long long first = 0; first < total; first += chunk) {
    const long long cnt = total - first < chunk ? total - first : chunk;
    hipLaunchKernelGGL(k_shard_degrees, dim3(8192), dim3(threads), 0, stream, scale, first, cnt, seed, deg.as<int>());
  }
  hipLaunchKernelGGL(k_iota, dim3(nb), dim3(threads), 0, stream, n, ids.as<int>());
  size_t bytes = 0;
  tmp_t scratch;
  SHARD_TRY(rocprim::radix_sort_pairs_desc(nullptr, bytes, deg.as<int>(), P->deg_sorted.as<int>(), ids.as<int>(), P->old_of_new.as<int>(),
                                           (size_t)n, 0, 32, stream));
  SHARD_TRY(scratch.alloc(bytes));
  SHARD_TRY(rocprim::radix_sort_pairs_desc(scratch.p, bytes, deg.as<int>(), P->deg_sorted.as<int>(), ids.as<int>(), P->old_of_new.as<int>(),
                                           (size_t)n, 0, 32, stream));
  hipLaunchKernelGGL(k_invert, dim3(nb), dim3(threads), 0, stream, P->old_of_new.as<int>(), n, P->new_of_old.as<int>());
  // local row offsets: exclusive scan of the owned vertices' degrees
  tmp_t ldeg, st;
  SHARD_TRY(ldeg.alloc(((size_t)P->n_local + 1) * 4)); SHARD_TRY(P->ro_local.alloc(((size_t)P->n_local + 1) * 4));
  SHARD_TRY(hipMemsetAsync(ldeg.p, 0, ((size_t)P->n_local + 1) * 4, stream));
  hipLaunchKernelGGL(k_shard_local_degrees, dim3((unsigned)(((long long)P->n_local + threads - 1) / threads)), dim3(threads), 0, stream,
                     P->deg_sorted.as<int>(), n, ranks, rank, P->n_local, ldeg.as<int>());
  size_t sb = 0;
  SHARD_TRY(rocprim::exclusive_scan(nullptr, sb, ldeg.as<int>(), P->ro_local.as<int>(), 0, (size_t)P->n_local + 1, rocprim::plus<int>(), stream));
  SHARD_TRY(st.alloc(sb));
  SHARD_TRY(rocprim::exclusive_scan(st.p, sb, ldeg.as<int>(), P->ro_local.as<int>(), 0, (size_t)P->n_local + 1, rocprim::plus<int>(), stream));
  // the shard's entry count in 64 bits: the int32 scan above wraps for a shard of 2^31 entries or more, and a wrapped (negative
  // or small) total would pass the caller's range check and come back as an empty or truncated shard
  tmp_t total64, rt;
  SHARD_TRY(total64.alloc(8));
  size_t rb = 0;
  SHARD_TRY(rocprim::reduce(nullptr, rb, ldeg.as<int>(), total64.as<long long>(), 0ll, (size_t)P->n_local, rocprim::plus<long long>(), stream));
  SHARD_TRY(rt.alloc(rb));
  SHARD_TRY(rocprim::reduce(rt.p, rb, ldeg.as<int>(), total64.as<long long>(), 0ll, (size_t)P->n_local, rocprim::plus<long long>(), stream));
  long long last = 0;
  SHARD_TRY(hipMemcpyAsync(&last, total64.p, 8, hipMemcpyDeviceToHost, stream));
  SHARD_TRY(hipStreamSynchronize(stream));
  P->m_local = last;
  *handle = P; *n_local = P->n_local; *m_local = P->m_local;
  return 0;
}

// fill: row_offsets (n_local + 1), col (m_local), new_of_old / old_of_new / deg_of_new (n each; any may be NULL) -- device buffers
extern "C" int mgx_shard_fill_device(void* handle, int* row_offsets, int* col, int* new_of_old, int* old_of_new, int* deg_of_new,
                                     hipStream_t stream) {
  shard_plan_t* P = (shard_plan_t*)handle;
  if (!P) return (int)hipErrorInvalidValue;
  const int n = P->n;
  LAY_TRY(hipMemcpyAsync(row_offsets, P->ro_local.p, ((size_t)P->n_local + 1) * 4, hipMemcpyDeviceToDevice, stream));
  if (new_of_old) LAY_TRY(hipMemcpyAsync(new_of_old, P->new_of_old.p, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
  if (old_of_new) LAY_TRY(hipMemcpyAsync(old_of_new, P->old_of_new.p, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
  if (deg_of_new) LAY_TRY(hipMemcpyAsync(deg_of_new, P->deg_sorted.p, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
  const long long m = P->m_local;
  if (m <= 0) { LAY_TRY(hipStreamSynchronize(stream)); return 0; }
  tmp_t keys, keys_sorted, cursor, scratch;
  LAY_TRY(keys.alloc((size_t)m * 8)); LAY_TRY(keys_sorted.alloc((size_t)m * 8)); LAY_TRY(cursor.alloc(8));
  LAY_TRY(hipMemsetAsync(cursor.p, 0, 8, stream));
  const long long total = (long long)P->edgefactor * n;
  const long long chunk = 1ll << 28;
  for (long long first = 0; first < total; first += chunk) {
    const long long cnt = total - first < chunk ? total - first : chunk;
    hipLaunchKernelGGL(k_shard_keys, dim3(8192), dim3(256), 0, stream, P->scale, first, cnt, P->seed, P->new_of_old.as<int>(), P->ranks,
                       P->rank, keys.as<unsigned long long>(), cursor.as<unsigned long long>(), (unsigned long long)m);
  }
  unsigned long long filled = 0;
  LAY_TRY(hipMemcpyAsync(&filled, cursor.p, 8, hipMemcpyDeviceToHost, stream));
  LAY_TRY(hipStreamSynchronize(stream));
  if ((long long)filled != m) return (int)hipErrorUnknown;        // (the two passes must agree: same stream, same permutation)
  int row_bits = 1;
  while (row_bits < 31 && (1ll << row_bits) < (long long)P->n_local) ++row_bits;
  size_t bytes = 0;
  LAY_TRY(rocprim::radix_sort_keys(nullptr, bytes, keys.as<unsigned long long>(), keys_sorted.as<unsigned long long>(), (size_t)m, 0,
                                   32 + row_bits, stream));
  LAY_TRY(scratch.alloc(bytes));
  LAY_TRY(rocprim::radix_sort_keys(scratch.p, bytes, keys.as<unsigned long long>(), keys_sorted.as<unsigned long long>(), (size_t)m, 0,
                                   32 + row_bits, stream));
  hipLaunchKernelGGL(k_shard_split, dim3(8192), dim3(256), 0, stream, keys_sorted.as<unsigned long long>(), m, col);
  LAY_TRY(hipStreamSynchronize(stream));
  return 0;
}
extern "C" void mgx_shard_free_device(void* handle) { delete (shard_plan_t*)handle; }

// ---- the long rows by SLICE of their destinations, for the neighbour-reduce (mgx/nreduce.hpp: k_nrs_edges) ----------------
// The full-frontier reduce gathers value(dst) per entry; only slice_n values (160 KB) fit the LDS of a compute unit, and on
// RMAT-22 the first 40 000 layout vertices are 60 % of the long rows' endpoints -- the other 40 % were 4-byte gathers through
// the L2, what bounded the kernel.  Here the entries of the rows [0, rows) are regrouped slice-major: slice k < slices holds
// the entries with dst in [k * slice_n, (k + 1) * slice_n), slice `slices` (the TAIL) everything behind.  Inside a slice the
// rows follow each other, each row's entries (the rows are sorted: a contiguous piece of the row) cut into MINI-UNITS of 16
// bytes -- 8 offsets into the slice at 16 bits each (padding: slice_n, where the kernel keeps the identity), in the tail 4
// layout ids at 32 bits (padding: -1).  A workgroup that takes mini-units of ONE slice keeps that slice's values in LDS: no
// gather leaves the compute unit, one lane folds one mini-unit, and off[k * rows + r] says where row r's partials of slice k
// start.  RMAT-22: 16 slices cover 94 % of the long rows' entries; 17.5 M mini-units, 1.21 slots per entry.
namespace {

// the entries of row [r0, r1) below `bound` (rows are sorted by neighbour)
__device__ __forceinline__ int nrs_lower(const int* __restrict__ ci, int r0, int r1, unsigned long long bound) {
  int lo = r0, hi = r1;
  while (lo < hi) { const int mid = lo + (hi - lo) / 2; if ((unsigned long long)(unsigned)ci[mid] < bound) lo = mid + 1; else hi = mid; }
  return lo;
}

// cnt[k * rows + r] = mini-units of (slice k, row r); cnt[(slices + 1) * rows] = 0 (so that the scan's last entry is the total)
__global__ void k_nrs_counts(const int* __restrict__ ro, const int* __restrict__ ci, int rows, unsigned slice_n, int slices,
                             unsigned* __restrict__ cnt) {
  const long long cells = (long long)(slices + 1) * rows;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == cells) cnt[i] = 0u;
  if (i >= cells) return;
  const int k = (int)(i / rows), r = (int)(i % rows);
  const int r0 = ro[r], r1 = ro[r + 1];
  const int a = k == 0 ? r0 : nrs_lower(ci, r0, r1, (unsigned long long)k * slice_n);
  const int b = k == slices ? r1 : nrs_lower(ci, r0, r1, (unsigned long long)(k + 1) * slice_n);
  const unsigned c = (unsigned)(b - a);
  cnt[i] = k < slices ? (c + 7u) / 8u : (c + 3u) / 4u;
}

// mini-unit j: its cell by a search in the scanned counts (setup code), its entries from the row
__global__ void k_nrs_fill(const int* __restrict__ ro, const int* __restrict__ ci, int rows, unsigned slice_n, int slices,
                           const unsigned* __restrict__ off, unsigned total, uint4* __restrict__ mu) {
  const long long cells = (long long)(slices + 1) * rows;
  for (unsigned j = blockIdx.x * blockDim.x + threadIdx.x; j < total; j += gridDim.x * blockDim.x) {
    long long lo = 0, hi = cells;                   // last cell with off[cell] <= j (it is not empty: off[cell + 1] > j)
    while (hi - lo > 1) { const long long mid = lo + (hi - lo) / 2; if (off[mid] <= j) lo = mid; else hi = mid; }
    const int k = (int)(lo / rows), r = (int)(lo % rows);
    const unsigned t = j - off[lo];
    const int r0 = ro[r], r1 = ro[r + 1];
    const int a = k == 0 ? r0 : nrs_lower(ci, r0, r1, (unsigned long long)k * slice_n);
    const int b = k == slices ? r1 : nrs_lower(ci, r0, r1, (unsigned long long)(k + 1) * slice_n);
    unsigned w[4];
    if (k < slices) {
      const unsigned base = (unsigned)k * slice_n;
      unsigned e[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { const int at = a + (int)(t * 8u) + q; e[q] = at < b ? (unsigned)ci[at] - base : slice_n; }
#pragma unroll
      for (int q = 0; q < 4; ++q) w[q] = e[2 * q] | (e[2 * q + 1] << 16);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int at = a + (int)(t * 4u) + q; w[q] = at < b ? (unsigned)ci[at] : 0xFFFFFFFFu; }
    }
    mu[j] = make_uint4(w[0], w[1], w[2], w[3]);
  }
}

}  // namespace

// The rows [0, rows) of the layout (ro / ci: device, rows sorted by neighbour), slice_n <= 65535 values per slice, `slices` hot
// slices + the tail.  Allocates *mu (16 bytes per mini-unit, + 64 of slack) and *off ((slices + 1) * rows + 1 words) with hipMalloc;
// first[k] (host, slices + 2 words) = first mini-unit of slice k, first[slices + 1] = *total.  Nothing is allocated when
// there are no mini-units or more than 2^31 of them.  Returns with `stream` synchronised.
extern "C" int mgx_nrs_build_device(const int* ro, const int* ci, int rows, unsigned slice_n, int slices, void** mu, unsigned** off,
                                    unsigned* first, long long* total, hipStream_t stream) {
  *mu = nullptr; *off = nullptr; *total = 0;
  for (int k = 0; k <= slices + 1; ++k) first[k] = 0u;
  if (rows <= 0 || slices <= 0 || slice_n == 0u || slice_n > 65535u) return 0;
  const size_t cells = (size_t)(slices + 1) * (size_t)rows;
  tmp_t cnt, st;
  unsigned* offs = nullptr;
  LAY_TRY(cnt.alloc((cells + 1) * 4));
  LAY_TRY(hipMalloc((void**)&offs, (cells + 1) * 4));
  hipLaunchKernelGGL(k_nrs_counts, dim3((unsigned)((cells + 1 + 255) / 256)), dim3(256), 0, stream, ro, ci, rows, slice_n, slices,
                     cnt.as<unsigned>());
  // (64-bit sums would be needed past 2^32 mini-units: the entries are fewer than 2^31, a mini-unit holds at least one)
  size_t sb = 0;
  hipError_t e = rocprim::exclusive_scan(nullptr, sb, cnt.as<unsigned>(), offs, 0u, cells + 1, rocprim::plus<unsigned>(), stream);
  if (e == hipSuccess) e = st.alloc(sb);
  if (e == hipSuccess) e = rocprim::exclusive_scan(st.p, sb, cnt.as<unsigned>(), offs, 0u, cells + 1, rocprim::plus<unsigned>(), stream);
  for (int k = 0; k <= slices + 1 && e == hipSuccess; ++k)
    e = hipMemcpyAsync(first + k, offs + (size_t)k * rows, 4, hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  const unsigned M = first[slices + 1];
  if (e != hipSuccess || M == 0u || M >= (1u << 31)) { (void)hipFree(offs); for (int k = 0; k <= slices + 1; ++k) first[k] = 0u; return (int)e; }
  void* units = nullptr;
  e = hipMalloc(&units, ((size_t)M + 4) * 16);
  if (e != hipSuccess) { (void)hipFree(offs); for (int k = 0; k <= slices + 1; ++k) first[k] = 0u; (void)hipGetLastError(); return 0; }   // (no memory: no slices)
  hipLaunchKernelGGL(k_nrs_fill, dim3(8192), dim3(256), 0, stream, ro, ci, rows, slice_n, slices, (const unsigned*)offs, M, (uint4*)units);
  e = hipMemsetAsync((char*)units + (size_t)M * 16, 0xFF, 64, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  if (e != hipSuccess) { (void)hipFree(offs); (void)hipFree(units); for (int k = 0; k <= slices + 1; ++k) first[k] = 0u; return (int)e; }
  *mu = units; *off = offs; *total = (long long)M;
  return 0;
}
