// What a CU-masked stream does on this part: distinct compute units a grid lands on, and whether a masked small kernel runs
// beside a device-filling one.  hipcc --offload-arch=gfx950 -O2 tools/cumask_probe.hip -o tools/cumask_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <set>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_where(unsigned* out) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) out[blockIdx.x] = ((xcc & 0xF) << 16) | (hw & 0xFFFF);
  for (volatile int i = 0; i < 2000; ++i) {}
}
__global__ __launch_bounds__(1024) void k_busy(float* p, int iters) {
  extern __shared__ float lds[];
  float a = p[threadIdx.x];
  for (int i = 0; i < iters; ++i) a = a * 1.0001f + 0.5f;
  lds[threadIdx.x] = a; __syncthreads();
  p[blockIdx.x * 1024 + threadIdx.x] = lds[(threadIdx.x + 1) & 1023];
}
__global__ void k_small(float* p) { p[threadIdx.x] += 1.f; }
int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int ncu = pr.multiProcessorCount;
  printf("CUs %d\n", ncu);
  const int words = (ncu + 31) / 32;
  for (int reserve : {8, 16}) {
    std::vector<uint32_t> big(words, 0), small(words, 0);
    // reserve CUs spread over the mask's index space
    for (int i = 0; i < ncu; ++i) {
      const bool r = (i % (ncu / reserve)) == 0 && (i / (ncu / reserve)) < reserve;
      (r ? small : big)[i / 32] |= 1u << (i % 32);
    }
    hipStream_t sb, ss;
    CK(hipExtStreamCreateWithCUMask(&sb, words, big.data()));
    CK(hipExtStreamCreateWithCUMask(&ss, words, small.data()));
    unsigned* d; CK(hipMalloc(&d, 8192 * 4));
    std::vector<unsigned> h(8192);
    for (int which = 0; which < 2; ++which) {
      hipStream_t s = which ? ss : sb;
      hipLaunchKernelGGL(k_where, dim3(8192), dim3(64), 0, s, d);
      CK(hipStreamSynchronize(s));
      CK(hipMemcpy(h.data(), d, 8192 * 4, hipMemcpyDeviceToHost));
      std::set<unsigned> cus, xccs;
      for (unsigned v : h) { cus.insert(((v >> 16) << 16) | (v & 0xFF00) | ((v >> 8) & 0)); xccs.insert(v >> 16); }
      std::set<unsigned> ids;
      for (unsigned v : h) ids.insert(((v >> 16) << 12) | ((v >> 8) & 0xF) /*cu*/ | (((v >> 12) & 0x1) << 4) /*sh*/ | (((v >> 13) & 0x7) << 5) /*se*/);
      printf("reserve %d, %s stream: %zu distinct (xcc, se, sh, cu), %zu xccs\n", reserve, which ? "small" : "big", ids.size(), xccs.size());
    }
    // a device-filling kernel on the big stream (2 x 1024 threads x 80 KB per CU) and small kernels beside it on the small one
    float* p; CK(hipMalloc(&p, (size_t)4096 * 1024 * 4)); CK(hipMemset(p, 0, (size_t)4096 * 1024 * 4));
    CK(hipFuncSetAttribute((const void*)k_busy, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    hipEvent_t e0, e1, f0, f1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&f0)); CK(hipEventCreate(&f1));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0, sb));
      hipLaunchKernelGGL(k_busy, dim3(2 * (ncu - reserve)), dim3(1024), 80 * 1024, sb, p, 200000);
      CK(hipEventRecord(e1, sb));
      CK(hipEventRecord(f0, ss));
      for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_small, dim3(1), dim3(256), 0, ss, p + 4000 * 1024);
      CK(hipEventRecord(f1, ss));
      CK(hipDeviceSynchronize());
      float tb, ts; CK(hipEventElapsedTime(&tb, e0, e1)); CK(hipEventElapsedTime(&ts, f0, f1));
      printf("reserve %d: busy kernel %.1f us; 20 small kernels on the small stream meanwhile %.1f us\n", reserve, tb * 1e3, ts * 1e3);
    }
    // the same small kernels on an UNMASKED second stream beside a full-device busy kernel
    hipStream_t su, sv; CK(hipStreamCreateWithFlags(&su, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sv, hipStreamNonBlocking));
    CK(hipEventRecord(e0, su));
    hipLaunchKernelGGL(k_busy, dim3(2 * ncu), dim3(1024), 80 * 1024, su, p, 200000);
    CK(hipEventRecord(e1, su));
    CK(hipEventRecord(f0, sv));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k_small, dim3(1), dim3(256), 0, sv, p + 4000 * 1024);
    CK(hipEventRecord(f1, sv));
    CK(hipDeviceSynchronize());
    float tb, ts; CK(hipEventElapsedTime(&tb, e0, e1)); CK(hipEventElapsedTime(&ts, f0, f1));
    printf("unmasked: busy kernel %.1f us; 20 small kernels on another stream meanwhile %.1f us\n", tb * 1e3, ts * 1e3);
  }
  return 0;
}
