// tools/microbench2.hip -- how fast can short random rows of a big array be streamed?
// Models the col_indices reads of a BFS level: R rows of `len` ints at given start offsets, one wave
// per chunk of rows, D independent 256-byte (64 lanes x 4 B) reads in flight per wave.
// Compares row order: random (discovery order) vs ascending address (sorted frontier).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int D>
__global__ void k_rows(const unsigned* __restrict__ data, const unsigned* __restrict__ starts, int nrows, int len,
                       unsigned* out) {
  const int lane = threadIdx.x & 63;
  const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int nwaves = (gridDim.x * blockDim.x) >> 6;
  const int per = (nrows + nwaves - 1) / nwaves;
  const int r0 = wave * per, r1 = min(nrows, r0 + per);
  const int cpr = (len + 63) / 64;             // 256-byte reads per row
  unsigned acc = 0;
  // flatten (row, chunk) pairs; D reads in flight
  const long long total = (long long)(r1 - r0) * cpr;
  for (long long t = 0; t < total; t += D) {
    unsigned v[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
      const long long tt = (t + d < total) ? t + d : total - 1;
      const int r = r0 + (int)(tt / cpr);
      const int c = (int)(tt % cpr);
      const unsigned off = c * 64 + lane;
      v[d] = data[starts[r] + (off < (unsigned)len ? off : 0u)];
    }
#pragma unroll
    for (int d = 0; d < D; ++d) acc ^= v[d];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <typename F>
static float time_ms(F f, int reps = 5) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const size_t N = (size_t)134217728;          // ints = 512 MB, like RMAT-22 col_indices
  unsigned *data, *out; CK(hipMalloc(&data, N * 4)); CK(hipMemset(data, 1, N * 4)); CK(hipMalloc(&out, 64));
  std::mt19937_64 rng(1);
  for (int len : {16, 90, 256, 1024, 4096}) {
    const int nrows = (int)std::min<size_t>(115000000 / len, 4000000);
    std::vector<unsigned> st(nrows);
    for (auto& s : st) s = (unsigned)(rng() % (N - len - 64));
    std::vector<unsigned> sorted = st; std::sort(sorted.begin(), sorted.end());
    unsigned* d_st; CK(hipMalloc(&d_st, nrows * 4));
    for (int order = 0; order < 2; ++order) {
      CK(hipMemcpy(d_st, order ? sorted.data() : st.data(), nrows * 4, hipMemcpyHostToDevice));
      for (int wpc : {16, 32}) {
        const int grid = 256 * wpc / 4;        // 256-thread blocks
        const double bytes = (double)nrows * len * 4;
        float m1 = time_ms([&] { hipLaunchKernelGGL(k_rows<1>, dim3(grid), dim3(256), 0, 0, data, d_st, nrows, len, out); });
        float m4 = time_ms([&] { hipLaunchKernelGGL(k_rows<4>, dim3(grid), dim3(256), 0, 0, data, d_st, nrows, len, out); });
        float m8 = time_ms([&] { hipLaunchKernelGGL(k_rows<8>, dim3(grid), dim3(256), 0, 0, data, d_st, nrows, len, out); });
        float m16 = time_ms([&] { hipLaunchKernelGGL(k_rows<16>, dim3(grid), dim3(256), 0, 0, data, d_st, nrows, len, out); });
        printf("{\"bench\":\"rows\",\"len\":%d,\"rows\":%d,\"order\":\"%s\",\"waves_per_cu\":%d,\"GBps_D1\":%.0f,\"GBps_D4\":%.0f,\"GBps_D8\":%.0f,\"GBps_D16\":%.0f}\n",
               len, nrows, order ? "sorted" : "random", wpc, bytes / m1 / 1e6, bytes / m4 / 1e6, bytes / m8 / 1e6, bytes / m16 / 1e6);
      }
    }
    CK(hipFree(d_st));
  }
  return 0;
}
