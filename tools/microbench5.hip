// tools/microbench5.hip -- random 4-byte GATHERS on gfx950 by table size: what the fused SSSP's look at dist[dst] costs when the
// destinations of a pass are confined to a slice of the array.  256 workgroups of 1024 threads (one per CU, the sweep's shape), every lane
// four independent gathers per step at hashed indices below `mask + 1` words, with a 16-byte-per-lane nontemporal stream beside them
// (the unit blocks the sweep reads).  One JSON object per line.  Build: hipcc --offload-arch=gfx950 -O3 tools/microbench5.hip -o tools/microbench5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <bool STREAM>
__global__ __launch_bounds__(1024) void k_gather(const unsigned* __restrict__ table, unsigned mask, unsigned base_step, int per, const uint4* __restrict__ big,
                                                 size_t big16, unsigned* out) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t bi = tid;
  unsigned acc = 0;
  for (int it = 0; it < per; ++it) {
    if (STREAM) { const u32x4 v = __builtin_nontemporal_load((const u32x4*)big + (bi % big16)); acc ^= v.x; bi += stride; }
    // (base_step != 0: the window of the table moves every 64 steps -- all workgroups in the same slice at about the same time)
    const unsigned base = base_step ? (unsigned)(it >> 6) * base_step : 0u;
    unsigned g[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) g[k] = __hip_atomic_load(table + base + (hash32(tid * 977u + (unsigned)(it * 4 + k) * 0x9E3779B1u) & mask), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    acc ^= g[0] ^ g[1] ^ g[2] ^ g[3];
  }
  if (acc == 0x12345678u) out[0] = acc;
}

int main() {
  const size_t table_words = 64u << 20;            // 256 MB
  unsigned* table; CK(hipMalloc(&table, table_words * 4)); CK(hipMemset(table, 1, table_words * 4));
  const size_t big16 = (512u << 20) / 16;
  uint4* big; CK(hipMalloc(&big, big16 * 16)); CK(hipMemset(big, 0, big16 * 16));
  unsigned* out; CK(hipMalloc(&out, 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int per = 512;                              // steps per lane: 256 x 1024 x 512 x 4 = 537 M gathers per launch
  for (int stream = 0; stream < 2; ++stream)
    for (unsigned kb : {128u, 512u, 2048u, 4096u, 16384u, 65536u, 262144u}) {
      const unsigned mask = kb * 256u - 1u;         // words
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(e0));
        if (stream) hipLaunchKernelGGL(k_gather<true>, dim3(256), dim3(1024), 0, 0, table, mask, 0u, per, big, big16, out);
        else hipLaunchKernelGGL(k_gather<false>, dim3(256), dim3(1024), 0, 0, table, mask, 0u, per, big, big16, out);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double n = 256.0 * 1024.0 * per * 4.0;
        if (rep) printf("{\"bench\": \"gather4\", \"table_KB\": %u, \"stream\": %d, \"ms\": %.3f, \"G_gathers_per_s\": %.1f}\n", kb, stream, ms, n / ms / 1e6);
      }
    }
  // a 16 MB array walked in eight 2 MB windows (all workgroups move together) against the same array at random
  for (unsigned win_kb : {2048u, 4096u}) {
    const unsigned mask = win_kb * 256u - 1u, step = win_kb * 256u;
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_gather<true>, dim3(256), dim3(1024), 0, 0, table, mask, step, per, big, big16, out);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double n = 256.0 * 1024.0 * per * 4.0;
      if (rep) printf("{\"bench\": \"gather4_windows\", \"window_KB\": %u, \"windows\": %d, \"stream\": 1, \"ms\": %.3f, \"G_gathers_per_s\": %.1f}\n", win_kb, per / 64, ms, n / ms / 1e6);
    }
  }
  return 0;
}
