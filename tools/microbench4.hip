// tools/microbench4.hip -- scattered STORES on gfx950: what a "mark" costs.
// The push kernels of the fused BFS store one byte per (possibly) new neighbour; on the big level of RMAT-22 those
// stores cost 40-70 us of a 150-190 us launch (MGX_BFS_DENSE_DIAG=1).  This measures the store path alone:
//   width      1, 2, 4 bytes per store (sub-dword stores may need a read-modify-write under ECC)
//   table      512 KB ... 64 MB (L2 is 4 MB per XCD; eight XCDs each keep their own dirty copy of a line)
//   density    64 / 16 / 4 active lanes per wave instruction
//   with / without a concurrent 16-byte-per-lane stream (the col_indices reads the marks compete with)
// One JSON object per line.  Build: hipcc --offload-arch=gfx950 -O3 -Iinclude tools/microbench4.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

// every thread issues `per` stores of W bytes at hashed element indices; only lanes with (lane % LANEDIV) == 0 store;
// STREAM: between two stores the thread also reads 16 bytes of a big array (coalesced over the wave)
template <typename T, int LANEDIV, bool STREAM, bool NT>
__global__ __launch_bounds__(1024) void k_scatter(T* table, unsigned mask, int per, const uint4* __restrict__ big, size_t big16, unsigned* out) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = (threadIdx.x % LANEDIV) == 0;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t bi = tid;
  unsigned acc = 0;
  for (int it = 0; it < per; ++it) {
    if (STREAM) { const u32x4 v = __builtin_nontemporal_load((const u32x4*)big + (bi % big16)); acc ^= v.x ^ v.y ^ v.z ^ v.w; bi += stride; }
    const unsigned h = hash32(tid * 977u + it * 0x9E3779B1u) & mask;
    if (active) {
      if (NT) __builtin_nontemporal_store((T)1, table + h);
      else table[h] = (T)1;
    }
  }
  if (acc == 0x12345678u) out[0] = acc;
}

template <typename T, int LANEDIV, bool STREAM, bool NT>
static void run(const char* name, T* table, size_t bytes, const uint4* big, size_t big16, unsigned* out) {
  const unsigned mask = (unsigned)(bytes / sizeof(T)) - 1u;
  const int per = 64, blocks = 512;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_scatter<T, LANEDIV, STREAM, NT>), dim3(blocks), dim3(1024), 0, 0, table, mask, per, big, big16, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double stores = (double)blocks * 1024 / LANEDIV * per;
  printf("{\"bench\":\"scatter_store\",\"variant\":\"%s\",\"width\":%d,\"active_lanes\":%d,\"stream\":%d,\"nontemporal\":%d,\"table_bytes\":%zu,"
         "\"us\":%.1f,\"Gstores_per_s\":%.1f,\"stream_GBps\":%.0f}\n",
         name, (int)sizeof(T), 64 / LANEDIV, (int)STREAM, (int)NT, bytes, ms * 1e3, stores / ms / 1e6,
         STREAM ? (double)blocks * 1024 * per * 16 / ms / 1e6 : 0.0);
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main() {
  const size_t big_bytes = 1ull << 30;
  void *table, *big; unsigned* out;
  CK(hipMalloc(&table, 64u << 20)); CK(hipMalloc(&big, big_bytes)); CK(hipMalloc(&out, 64));
  CK(hipMemset(table, 0, 64u << 20)); CK(hipMemset(big, 1, big_bytes));
  const uint4* b = (const uint4*)big; const size_t b16 = big_bytes / 16;
  const size_t sizes[] = {512u << 10, 4u << 20, 16u << 20, 64u << 20};
  for (size_t sz : sizes) {
    run<unsigned char, 1, false, false>("byte", (unsigned char*)table, sz, b, b16, out);
    run<unsigned short, 1, false, false>("short", (unsigned short*)table, sz, b, b16, out);
    run<unsigned, 1, false, false>("dword", (unsigned*)table, sz, b, b16, out);
    run<unsigned char, 1, false, true>("byte_nt", (unsigned char*)table, sz, b, b16, out);
    run<unsigned char, 4, false, false>("byte_16lanes", (unsigned char*)table, sz, b, b16, out);
    run<unsigned char, 16, false, false>("byte_4lanes", (unsigned char*)table, sz, b, b16, out);
    run<unsigned, 16, false, false>("dword_4lanes", (unsigned*)table, sz, b, b16, out);
  }
  // the marks' situation: a 4 MB byte table under a 16-byte-per-lane stream, a store from 1/4 or 1/16 of the lanes per load
  run<unsigned char, 1, true, false>("byte+stream", (unsigned char*)table, 4u << 20, b, b16, out);
  run<unsigned char, 4, true, false>("byte_16lanes+stream", (unsigned char*)table, 4u << 20, b, b16, out);
  run<unsigned char, 16, true, false>("byte_4lanes+stream", (unsigned char*)table, 4u << 20, b, b16, out);
  run<unsigned, 4, true, false>("dword_16lanes+stream", (unsigned*)table, 16u << 20, b, b16, out);
  run<unsigned, 16, true, false>("dword_4lanes+stream", (unsigned*)table, 16u << 20, b, b16, out);
  run<unsigned char, 4, true, true>("byte_nt_16lanes+stream", (unsigned char*)table, 4u << 20, b, b16, out);
  run<unsigned char, 64, true, false>("byte_1lane+stream", (unsigned char*)table, 4u << 20, b, b16, out);
  // the stream alone (same loop, the stores compiled in but no lane active would change the code: a 1-lane store is the closest)
  return 0;
}
