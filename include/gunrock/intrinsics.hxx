// gunrock/intrinsics.hxx -- device helpers the functors call.
// Drop-in for the reference's gunrock/src/intrinsics.hxx:6-22.
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "../mgx/wave.hpp"

namespace gunrock {
namespace util {

__device__ __forceinline__ int LaneId() { return mgx::lane_id(); }

// float atomic min that returns the previous value (intrinsics.hxx:12-22 is a CAS loop).
// gfx950 has no native f32 min atomic, but IEEE-754 order can be had from the integer
// atomics: for val >= 0 a signed-int min on the bit pattern, for val < 0 an unsigned max.
// One global_atomic_smin / umax instead of a compare-and-swap retry loop; bit-identical
// result for every non-NaN input.
// (A template so that code which says `using namespace gunrock::util;` and then calls
// atomicMin(float*, float) unqualified -- the reference's sssp_functor.hxx:22 -- still resolves:
// HIP ships its own non-template ::atomicMin(float*, float), which then wins overload
// resolution instead of clashing; qualified calls gunrock::util::atomicMin(...) get this one.)
template <typename T>
__device__ __forceinline__ typename std::enable_if<std::is_same<T, float>::value, float>::type atomicMin(T* addr,
                                                                                                         T val) {
  if (val >= 0.0f) {
    int old = ::atomicMin((int*)addr, __float_as_int(val));
    return __int_as_float(old);
  }
  unsigned old = ::atomicMax((unsigned*)addr, __float_as_uint(val));
  return __uint_as_float(old);
}

}  // namespace util
}  // namespace gunrock
