// tools/microbench3.hip -- does it pay to keep loads in flight ACROSS rounds when streaming row-sized random chunks?
// Models the col_indices reads of k_bfs_push_level_stream: every wave reads "rows" of R consecutive 256-byte pieces
// (64 lanes x 4 B) at pseudo-random 4-byte-aligned places of a 512 MB array, 32 waves per CU.
//   drain<D>: issue D loads, wait for all of them, consume, repeat      (the shape of the kernel's round: the scalar
//             walk of the next round needs data that was loaded after the round's col_indices, so it drains)
//   ring<D>:  D loads always in flight: consume the oldest, issue one more (s_waitcnt vmcnt(D-1))
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ unsigned place(unsigned wave, unsigned piece, unsigned R, unsigned nmask) {
  const unsigned row = piece / R, in_row = piece - row * R;
  unsigned h = (wave * 0x9E3779B1u) ^ (row * 0x85EBCA77u);
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
  return ((h & nmask) + in_row * 64u);
}

template <int D, bool RING>
__global__ __launch_bounds__(1024, 8) void k_stream(const unsigned* __restrict__ data, unsigned pieces, unsigned R,
                                                     unsigned nmask, unsigned* out) {
  const unsigned lane = threadIdx.x & 63;
  const unsigned wave = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  unsigned acc = 0;
  unsigned v[D];
  if (RING) {
#pragma unroll
    for (int d = 0; d < D; ++d) v[d] = __builtin_nontemporal_load(data + place(wave, d, R, nmask) + lane);
    for (unsigned t = D; t < pieces; t += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) {
        acc ^= v[d];
        v[d] = __builtin_nontemporal_load(data + place(wave, t + d, R, nmask) + lane);
      }
    }
#pragma unroll
    for (int d = 0; d < D; ++d) acc ^= v[d];
  } else {
    for (unsigned t = 0; t < pieces; t += D) {
#pragma unroll
      for (int d = 0; d < D; ++d) v[d] = __builtin_nontemporal_load(data + place(wave, t + d, R, nmask) + lane);
#pragma unroll
      for (int d = 0; d < D; ++d) acc ^= v[d];
      asm volatile("" ::: "memory");
    }
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
  unsigned *data, *out; CK(hipMalloc(&data, (N + 65536) * 4)); CK(hipMemset(data, 1, (N + 65536) * 4)); CK(hipMalloc(&out, 64));
  const unsigned nmask = (unsigned)N - 1;
  const int grid = 512;                        // 2 workgroups of 1024 threads per CU: 32 waves per CU
  const unsigned pieces = 192;                 // per wave: 192 x 256 B; 8192 waves -> 403 MB per launch
  for (unsigned R : {1u, 4u, 16u, 192u}) {
    const double bytes = (double)grid * 16 * pieces * 256.0;
    float d8 = time_ms([&] { hipLaunchKernelGGL((k_stream<8, false>), dim3(grid), dim3(1024), 0, 0, data, pieces, R, nmask, out); });
    float d16 = time_ms([&] { hipLaunchKernelGGL((k_stream<16, false>), dim3(grid), dim3(1024), 0, 0, data, pieces, R, nmask, out); });
    float r8 = time_ms([&] { hipLaunchKernelGGL((k_stream<8, true>), dim3(grid), dim3(1024), 0, 0, data, pieces, R, nmask, out); });
    float r16 = time_ms([&] { hipLaunchKernelGGL((k_stream<16, true>), dim3(grid), dim3(1024), 0, 0, data, pieces, R, nmask, out); });
    float r24 = time_ms([&] { hipLaunchKernelGGL((k_stream<24, true>), dim3(grid), dim3(1024), 0, 0, data, pieces, R, nmask, out); });
    printf("{\"bench\":\"rows_inflight\",\"pieces_per_row\":%u,\"GBps_drain8\":%.0f,\"GBps_drain16\":%.0f,\"GBps_ring8\":%.0f,\"GBps_ring16\":%.0f,\"GBps_ring24\":%.0f}\n",
           R, bytes / d8 / 1e6, bytes / d16 / 1e6, bytes / r8 / 1e6, bytes / r16 / 1e6, bytes / r24 / 1e6);
  }
  return 0;
}
