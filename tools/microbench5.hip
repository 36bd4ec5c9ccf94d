// What do the counter adds at the end of a sweep kernel cost?  512 workgroups of 512 threads stream 4 MB each in all, then every
// workgroup adds to K 64-bit counters (no return value): all on ONE cache line, on K lines, or spread over 16 copies of the line
// by workgroup.  (k_bfs_build2's lazy path: reached, cursor, lcursor, ledges -- one line of the control block.)
// hipcc --offload-arch=gfx950 -O2 tools/microbench5.hip -o tools/microbench5
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(512) void k_sweep(const uint4* __restrict__ in, unsigned long long* ctr, int K, int line_stride, int copies, int returning) {
  uint4 v = in[(size_t)(blockIdx.x % 512) * 512 + threadIdx.x];
  unsigned s = v.x ^ v.y ^ v.z ^ v.w;
  s = __reduce_add_sync(~0ull, s);
  if (threadIdx.x == 0) {
    unsigned long long* base = ctr + (size_t)(blockIdx.x % copies) * 64;
    unsigned long long r = 0;
    for (int k = 0; k < K; ++k) {
      if (returning) r += atomicAdd(base + (size_t)k * line_stride, (unsigned long long)(s | 1u));
      else atomicAdd(base + (size_t)k * line_stride, (unsigned long long)(s | 1u));
    }
    if (returning && r == 0x123456789ull) ctr[4096] = r;
  }
}
int main() {
  const int G = 512;
  uint4* in; CK(hipMalloc(&in, (size_t)G * 512 * 16)); CK(hipMemset(in, 1, (size_t)G * 512 * 16));
  unsigned long long* ctr; CK(hipMalloc(&ctr, 8192 * 8)); CK(hipMemset(ctr, 0, 8192 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct { const char* what; int K, stride, copies, ret; } cases[] = {
    {"no adds", 0, 1, 1, 0}, {"1 add, one address", 1, 1, 1, 0}, {"4 adds, one line", 4, 1, 1, 0}, {"4 adds, 4 lines", 4, 16, 1, 0},
    {"4 adds, one line, 16 copies by workgroup", 4, 1, 16, 0}, {"4 adds, one line, 64 copies", 4, 1, 64, 0},
    {"2 RETURNING adds, one line", 2, 1, 1, 1}, {"2 returning adds, one line, 16 copies", 2, 1, 16, 1}};
  for (auto& c : cases) {
    for (int grid : {512, 2048}) {
      float best = 1e9f;
      for (int rep = 0; rep < 20; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_sweep, dim3(grid), dim3(512), 0, 0, in, ctr, c.K, c.stride, c.copies, c.ret);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
      }
      printf("{\"case\": \"%s\", \"workgroups\": %d, \"us\": %.2f}\n", c.what, grid, best * 1e3);
    }
  }
  return 0;
}
