// see meta.hxx: host stand-in for golden generation only ("device" memory is a std::vector)
#pragma once
#include "context.hxx"
namespace mgpu {
template <class T>
struct mem_t {
  std::vector<T> v;
  mem_t() {}
  mem_t(size_t n, context_t&) : v(n) {}
  T* data() { return v.data(); }
  const T* data() const { return v.data(); }
  size_t size() const { return v.size(); }
  void swap(mem_t& r) { v.swap(r.v); }
  // kcore_problem.hxx:44 hands a mem_t<int> to problem_t::GetDegrees(mem_t<float>&) (problem.hxx:23): upstream that only
  // compiles against a moderngpu we do not have.  Here the call is let through as a view of the same 4-byte cells; what it
  // writes (floats into int cells) is discarded -- the driver refills kcore_problem_t::degrees before it calls cpu().
  template <class U>
  operator mem_t<U>&() {
    static_assert(sizeof(U) == sizeof(T), "view of equal-sized cells only");
    return *reinterpret_cast<mem_t<U>*>(this);
  }
};
template <class T>
mem_t<T> to_mem(const std::vector<T>& h, context_t& c) {
  mem_t<T> m(h.size(), c);
  m.v = h;
  return m;
}
template <class T>
std::vector<T> from_mem(const mem_t<T>& m) {
  return m.v;
}
template <class T>
mem_t<T> fill(T x, size_t n, context_t& c) {
  mem_t<T> m(n, c);
  for (auto& e : m.v) e = x;
  return m;
}
template <class T>
int dtoh(std::vector<T>& d, const T* s, size_t n) {
  d.assign(s, s + n);
  return 0;
}
template <class F>
void transform(F f, int n, context_t&) {
  for (int i = 0; i < n; ++i) f(i);
}
}  // namespace mgpu
