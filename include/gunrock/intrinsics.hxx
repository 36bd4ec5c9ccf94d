// gunrock/intrinsics.hxx -- device helpers the functors call.
// Drop-in for the reference's gunrock/src/intrinsics.hxx:6-22.
#pragma once
#include <hip/hip_runtime.h>
#include "../mgx/wave.hpp"

namespace gunrock {
namespace util {

__device__ __forceinline__ int LaneId() { return mgx::lane_id(); }

// float atomic min that returns the previous value (intrinsics.hxx:12-22 is a CAS loop).
// gfx950 has no native f32 min atomic, but IEEE-754 order can be had from the integer
// atomics: for val >= 0 a signed-int min on the bit pattern, for val < 0 an unsigned max.
// One global_atomic_smin / umax instead of a compare-and-swap retry loop; bit-identical
// result for every non-NaN input.
__device__ __forceinline__ float atomicMin(float* addr, float val) {
  if (val >= 0.0f) {
    int old = ::atomicMin((int*)addr, __float_as_int(val));
    return __int_as_float(old);
  }
  unsigned old = ::atomicMax((unsigned*)addr, __float_as_uint(val));
  return __uint_as_float(old);
}

}  // namespace util
}  // namespace gunrock
