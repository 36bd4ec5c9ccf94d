// see meta.hxx: host stand-in for golden generation only
#pragma once
#include "meta.hxx"
namespace mgpu {
struct context_t {
  virtual ~context_t() {}
};
struct standard_context_t : context_t {
  explicit standard_context_t(bool = false) {}
};
}  // namespace mgpu
