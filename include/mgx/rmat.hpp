// mgx/rmat.hpp -- the R-MAT pair of edge index e (spec: oracle/oracle.c orc_rmat_edges): counter-based
// (splitmix64 of (seed, edge, level)), (a, b, c, d) = (.57, .19, .19, .05) as 32-bit fixed point, bijective id scramble.
// One definition for the generator kernel (mgx_capi.hip: k_rmat_edges) and the shard builder (mgx_layout.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace mgx {

__host__ __device__ __forceinline__ unsigned long long rmat_mix64(unsigned long long z) {
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ unsigned rmat_scramble(unsigned v, int scale, unsigned mask) {
  v = (v * 0x9E3779B1u + 0x7F4A7C15u) & mask;
  v = __brev(v) >> (32 - scale);
  v = (v * 0x85EBCA6Bu + 0xC2B2AE35u) & mask;
  return v;
}
constexpr unsigned long long RMAT_K0 = 0xD1B54A32D192ED03ull, RMAT_K1 = 0x8CB92BA72F3D8DD7ull, RMAT_K2 = 0xA24BAED4963EE407ull;

// pair number e of the stream (seed, scale): (s, d) in [0, 2^scale)
__device__ __forceinline__ void rmat_pair(int scale, unsigned long long seed, unsigned long long e, int do_scramble, unsigned& s_out,
                                          unsigned& d_out) {
  const unsigned A = 2448131358u, AB = 3264175144u, ABC = 4080218931u;
  const unsigned mask = (scale >= 32) ? 0xFFFFFFFFu : ((1u << scale) - 1u);
  const unsigned long long x = rmat_mix64(seed * RMAT_K0 + e);
  unsigned s = 0, d = 0;
  for (int l = 0; l < scale; ++l) {
    const unsigned u = (unsigned)(rmat_mix64(x + (unsigned long long)(l + 1) * RMAT_K1) >> 32);
    const unsigned sb = (u >= AB) ? 1u : 0u;
    const unsigned db = (u < A) ? 0u : (u < AB) ? 1u : (u < ABC) ? 0u : 1u;
    s = (s << 1) | sb;
    d = (d << 1) | db;
  }
  if (do_scramble) { s = rmat_scramble(s, scale, mask); d = rmat_scramble(d, scale, mask); }
  s_out = s; d_out = d;
}
__device__ __forceinline__ float rmat_weight(unsigned long long seed, unsigned long long e) {
  return (float)(rmat_mix64(seed * RMAT_K0 + e + RMAT_K2) % 64ull);
}

}  // namespace mgx
