// CU-masked streams, second look: where and WHEN the workgroups of a device-filling grid start under a mask.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <map>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ __launch_bounds__(1024) void k_busy(float* p, int iters, unsigned* where, unsigned long long* t0, unsigned long long* t1) {
  extern __shared__ float lds[];
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) { where[blockIdx.x] = ((xcc & 0xF) << 16) | (hw & 0xFFFF); t0[blockIdx.x] = wall_clock64(); }
  float a = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
  lds[threadIdx.x] = a; __syncthreads();
  p[blockIdx.x * 1024 + threadIdx.x] = lds[(threadIdx.x + 1) & 1023];
  if (threadIdx.x == 0) t1[blockIdx.x] = wall_clock64();
}
static unsigned cu_of(unsigned v) { return ((v >> 16) << 12) | ((v >> 8) & 0xF) | (((v >> 12) & 0x1) << 4) | (((v >> 13) & 0x7) << 5); }
int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int ncu = pr.multiProcessorCount, words = (ncu + 31) / 32;
  float* p; CK(hipMalloc(&p, (size_t)4096 * 1024 * 4)); CK(hipMemset(p, 0, (size_t)4096 * 1024 * 4));
  unsigned* d; unsigned long long *t0, *t1;
  CK(hipMalloc(&d, 4096 * 4)); CK(hipMalloc(&t0, 4096 * 8)); CK(hipMalloc(&t1, 4096 * 8));
  CK(hipFuncSetAttribute((const void*)k_busy, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
  auto run = [&](const char* what, hipStream_t s, int grid) -> int {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(k_busy, dim3(grid), dim3(1024), 80 * 1024, s, p, 20000, d, t0, t1);
      CK(hipEventRecord(e1, s));
      CK(hipStreamSynchronize(s));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<unsigned> h(grid); std::vector<unsigned long long> a(grid), b(grid);
    CK(hipMemcpy(h.data(), d, grid * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(a.data(), t0, grid * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), t1, grid * 8, hipMemcpyDeviceToHost));
    std::map<unsigned, int> per; unsigned long long first = ~0ull, lastend = 0; double dur = 0;
    for (int i = 0; i < grid; ++i) { per[cu_of(h[i])]++; if (a[i] < first) first = a[i]; if (b[i] > lastend) lastend = b[i]; dur += (double)(b[i] - a[i]); }
    int late = 0; std::map<int, int> hist;
    for (int i = 0; i < grid; ++i) if ((a[i] - first) > (unsigned long long)(0.25 * dur / grid)) ++late;
    for (auto& kv : per) hist[kv.second]++;
    printf("%-34s grid %4d: %.1f us; %zu CUs; workgroups per CU:", what, grid, ms * 1e3, per.size());
    for (auto& kv : hist) printf(" %dx%d", kv.second, kv.first);
    printf("; started late (> a quarter of a workgroup's run): %d; mean workgroup run %.1f us, span %.1f us\n", late, dur / grid / 100.0, (lastend - first) / 100.0);
    return 0;
  };
  hipStream_t su; CK(hipStreamCreateWithFlags(&su, hipStreamNonBlocking));
  run("unmasked", su, 2 * ncu);
  run("unmasked", su, 2 * (ncu - 8));
  for (int pattern = 0; pattern < 4; ++pattern) {
    std::vector<uint32_t> big(words, 0);
    int kept = 0;
    for (int i = 0; i < ncu; ++i) {
      bool r = false;
      if (pattern == 0) r = (i % 32) == 0;            // first CU of every 32
      if (pattern == 1) r = (i % 32) == 31;           // last CU of every 32
      if (pattern == 2) r = i < 8;                    // the first eight
      if (pattern == 3) r = i >= ncu - 8;             // the last eight
      if (!r) { big[i / 32] |= 1u << (i % 32); ++kept; }
    }
    hipStream_t sb; CK(hipExtStreamCreateWithCUMask(&sb, words, big.data()));
    char name[64]; snprintf(name, sizeof name, "masked pattern %d (%d CUs kept)", pattern, kept);
    run(name, sb, 2 * kept);
    run(name, sb, 2 * ncu);
  }
  return 0;
}
