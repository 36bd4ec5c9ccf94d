// mgx_layout.hip -- hub-first layout of a device CSR: vertices renumbered by descending degree (stable:
// ties keep their original order), rows rebuilt under the new numbering with their neighbour lists sorted.
// Graph construction, not the hot path: one-time setup per graph, so it leans on rocPRIM's device-wide sort / scan
// instead of kernels of its own.  Same result as mini_amd.rmat.degree_order (torch device ops), which the tests
// compare it with.  A translation unit of its own: rocPRIM's headers are heavy, and nothing else needs them.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
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
