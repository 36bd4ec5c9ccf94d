// tools/microbench.hip -- gfx950 micro-measurements that size the BFS kernel design.
// Prints one JSON object per line.  Build: hipcc --offload-arch=gfx950 -O3 -Iinclude tools/microbench.hip
//   1. streaming read bandwidth (the col_indices stream)
//   2. random 4-byte gather rate vs table size (visited bitmap 512 KB ... label array 16-64 MB)
//   3. random device-scope atomicOr rate (claiming a vertex), returning and not
//   4. random workgroup-scope (L2-resident) atomicOr rate -- the per-XCD dedup idea
//   5. blockIdx -> XCC_ID placement
//   6. returning atomicAdd on ONE word (tile / queue counters)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "mgx/wave.hpp"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

__global__ void k_stream(const uint4* __restrict__ in, size_t n16, unsigned* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned acc = 0;
  for (; i < n16; i += stride) { uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) out[0] = acc;
}
__global__ void k_stream4(const unsigned* __restrict__ in, size_t n, unsigned* out) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned acc = 0;
  for (; i < n; i += stride) acc ^= in[i];
  if (acc == 0x12345678u) out[0] = acc;
}

// each thread does `per` gathers; indices are hashed (uniform over the table)
template <int UNROLL>
__global__ void k_gather(const unsigned* __restrict__ table, unsigned mask, int per, unsigned* out) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned acc = 0;
  for (int it = 0; it < per; it += UNROLL) {
    unsigned v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) v[u] = table[hash32(tid * 977u + (it + u) * 0x9E3779B1u) & mask];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) acc ^= v[u];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

// mode 0: device-scope atomicOr returning; 1: device-scope no-return; 2: workgroup-scope returning
// 3: test-then-device-atomic with bits mostly already set (the steady state of a BFS level)
template <int MODE>
__global__ void k_atomic(unsigned* table, unsigned mask, int per, unsigned* out) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned acc = 0;
  for (int it = 0; it < per; ++it) {
    const unsigned h = hash32(tid * 977u + it * 0x9E3779B1u);
    unsigned* p = table + (h & mask);
    const unsigned bit = 1u << (h >> 27);
    if (MODE == 0) acc ^= atomicOr(p, bit);
    else if (MODE == 1) atomicOr(p, bit);
    else if (MODE == 2) acc ^= __hip_atomic_fetch_or(p, bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else { if (!(*p & bit)) acc ^= atomicOr(p, bit); }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

__global__ void k_xcc(int* xcc_of_block) {
  if (threadIdx.x == 0) xcc_of_block[blockIdx.x] = mgx::xcc_id();
}

__global__ void k_counter(unsigned long long* ctr, int per, unsigned* out) {
  unsigned long long acc = 0;
  if (threadIdx.x == 0)
    for (int i = 0; i < per; ++i) acc ^= atomicAdd(ctr, 1ull);
  if (acc == 0x123456789ull) out[0] = 1;
}

template <typename F>
static float time_ms(F f, int reps = 5) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f();  // warm
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    CK(hipEventRecord(a));
    f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("{\"bench\":\"device\",\"name\":\"%s\",\"cus\":%d,\"clock_mhz\":%d,\"l2_bytes\":%d}\n", prop.name,
         prop.multiProcessorCount, prop.clockRate / 1000, prop.l2CacheSize);
  unsigned* out; CK(hipMalloc(&out, 64));
  const int grid = prop.multiProcessorCount * 8, block = 256;

  {  // 1. streaming
    const size_t bytes = (size_t)1 << 30;
    void* buf; CK(hipMalloc(&buf, bytes)); CK(hipMemset(buf, 1, bytes));
    float ms = time_ms([&] { hipLaunchKernelGGL(k_stream, dim3(grid), dim3(block), 0, 0, (const uint4*)buf, bytes / 16, out); });
    printf("{\"bench\":\"stream_read_16B\",\"GBps\":%.1f}\n", bytes / ms / 1e6);
    ms = time_ms([&] { hipLaunchKernelGGL(k_stream4, dim3(grid), dim3(block), 0, 0, (const unsigned*)buf, bytes / 4, out); });
    printf("{\"bench\":\"stream_read_4B\",\"GBps\":%.1f}\n", bytes / ms / 1e6);
    CK(hipFree(buf));
  }
  {  // 2. gathers
    const size_t sizes[] = {(size_t)32 << 10, (size_t)512 << 10, (size_t)4 << 20, (size_t)16 << 20, (size_t)64 << 20, (size_t)512 << 20};
    for (size_t sz : sizes) {
      unsigned* t; CK(hipMalloc(&t, sz)); CK(hipMemset(t, 0, sz));
      const unsigned mask = (unsigned)(sz / 4 - 1);
      const int per = 256;
      float ms = time_ms([&] { hipLaunchKernelGGL(k_gather<8>, dim3(grid), dim3(block), 0, 0, t, mask, per, out); });
      const double n = (double)grid * block * per;
      printf("{\"bench\":\"gather4B\",\"table_bytes\":%zu,\"Ggather_per_s\":%.2f}\n", sz, n / ms / 1e6);
      CK(hipFree(t));
    }
  }
  {  // 3/4. atomics
    const size_t sizes[] = {(size_t)512 << 10, (size_t)16 << 20};
    for (size_t sz : sizes) {
      unsigned* t; CK(hipMalloc(&t, sz));
      const unsigned mask = (unsigned)(sz / 4 - 1);
      const int per = 32;
      const double n = (double)grid * block * per;
      float ms;
      CK(hipMemset(t, 0, sz));
      ms = time_ms([&] { hipLaunchKernelGGL(k_atomic<0>, dim3(grid), dim3(block), 0, 0, t, mask, per, out); });
      printf("{\"bench\":\"atomicOr_device_ret\",\"table_bytes\":%zu,\"Gops_per_s\":%.2f}\n", sz, n / ms / 1e6);
      ms = time_ms([&] { hipLaunchKernelGGL(k_atomic<1>, dim3(grid), dim3(block), 0, 0, t, mask, per, out); });
      printf("{\"bench\":\"atomicOr_device_noret\",\"table_bytes\":%zu,\"Gops_per_s\":%.2f}\n", sz, n / ms / 1e6);
      ms = time_ms([&] { hipLaunchKernelGGL(k_atomic<2>, dim3(grid), dim3(block), 0, 0, t, mask, per, out); });
      printf("{\"bench\":\"atomicOr_workgroup_ret\",\"table_bytes\":%zu,\"Gops_per_s\":%.2f}\n", sz, n / ms / 1e6);
      CK(hipMemset(t, 0xFF, sz));
      ms = time_ms([&] { hipLaunchKernelGGL(k_atomic<3>, dim3(grid), dim3(block), 0, 0, t, mask, per, out); });
      printf("{\"bench\":\"test_then_atomic_allset\",\"table_bytes\":%zu,\"Gops_per_s\":%.2f}\n", sz, n / ms / 1e6);
      CK(hipFree(t));
    }
  }
  {  // 5. placement
    const int nb = 64;
    int* d; CK(hipMalloc(&d, nb * sizeof(int)));
    hipLaunchKernelGGL(k_xcc, dim3(nb), dim3(64), 0, 0, d);
    std::vector<int> h(nb);
    CK(hipMemcpy(h.data(), d, nb * sizeof(int), hipMemcpyDeviceToHost));
    printf("{\"bench\":\"xcc_of_block\",\"ids\":[");
    for (int i = 0; i < nb; ++i) printf("%d%s", h[i], i + 1 < nb ? "," : "");
    printf("]}\n");
    CK(hipFree(d));
  }
  {  // 6. one hot counter
    unsigned long long* c; CK(hipMalloc(&c, 8)); CK(hipMemset(c, 0, 8));
    const int per = 64;
    float ms = time_ms([&] { hipLaunchKernelGGL(k_counter, dim3(grid), dim3(64), 0, 0, c, per, out); });
    printf("{\"bench\":\"hot_counter_ret_atomicAdd\",\"Mops_per_s\":%.1f}\n", (double)grid * per / ms / 1e3);
    CK(hipFree(c));
  }
  return 0;
}
