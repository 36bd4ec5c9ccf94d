// gunrock/frontier.hxx -- frontier_t<T>: a device buffer of ids with a fill level.
// Drop-in for the reference's gunrock/src/frontier.hxx:12-99: the same public surface (load from a mem_t or a host
// vector, resize, capacity, size, type, data, swap, move-only).  Two deliberate differences: a frontier that would
// overflow throws mgx::mgx_error(MGX_E_FRONTIER_OVERFLOW) where the reference printf()s and exit(0)s (:53-59,
// :84-89); and constructing one reserves the context's scratch arena for a scan / compaction / segmented reduce over
// `capacity` items, so that the operators never allocate.
#pragma once

#include "graph.hxx"
#include "../mgx/scan.hpp"
#include "../mgx/lbs.hpp"

namespace gunrock {

enum frontier_type_t { edge_frontier = 0, node_frontier = 1 };

template <typename type_t>
class frontier_t {
 public:
  typedef std::shared_ptr<mem_t<type_t>> storage_t;

  frontier_t() : store_(std::make_shared<mem_t<type_t>>()) {}
  frontier_t(context_t& context, size_t capacity, size_t size = 0, frontier_type_t type = node_frontier)
      : store_(new mem_t<type_t>(capacity, context)), fill_(size), room_(capacity), kind_(type) {
    if (standard_context_t* ctx = dynamic_cast<standard_context_t*>(&context)) {
      const long long items = (long long)capacity;
      ctx->reserve_scratch(mgx::scan_scratch_bytes(items) + mgx::segreduce_scratch_bytes(items, 8));
    }
  }
  frontier_t(frontier_t&& other) : frontier_t() { swap(other); }
  frontier_t& operator=(frontier_t&& other) { swap(other); return *this; }
  frontier_t(const frontier_t&) = delete;
  frontier_t& operator=(const frontier_t&) = delete;

  void swap(frontier_t& other) {
    store_.swap(other.store_);
    std::swap(fill_, other.fill_);
    std::swap(room_, other.room_);
    std::swap(kind_, other.kind_);
    std::swap(exposed_, other.exposed_);
  }

  // contents <- a device array / a host vector (the fill level follows)
  hipError_t load(mem_t<type_t>& source) {
    admit("loading", source.size(), store_->size());
    fill_ = source.size();
    mgx::frontier_touched();
    return dtod(store_->data(), source.data(), source.size());
  }
  hipError_t load(const std::vector<type_t>& source) {
    admit("loading", source.size(), store_->size());
    fill_ = source.size();
    mgx::frontier_touched();
    return htod(store_->data(), source);
  }
  // an operator has written `count` entries
  void resize(size_t count) {
    admit("resizing", count, room_);
    fill_ = count;
  }

  size_t size() const { return fill_; }
  size_t capacity() const { return room_; }
  frontier_type_t type() const { return kind_; }
  storage_t data() const { return store_; }
  // the buffer's address has been handed to code outside the operators (the C-ABI's mgx_frontier_device_ptr): its contents may
  // change at any time from now on, so nothing an operator remembered about them is trusted again (filter.hxx)
  void mark_exposed() { exposed_ = true; mgx::frontier_touched(); }
  bool exposed() const { return exposed_; }

 private:
  static void admit(const char* doing, size_t want, size_t have) {
    if (want > have)
      throw mgx::mgx_error(MGX_E_FRONTIER_OVERFLOW, std::string("Overflow during frontier ") + doing + ". Capacity is " +
                                                        std::to_string(have) + ", size of the data is " +
                                                        std::to_string(want) + ".");
  }

  storage_t store_;
  size_t fill_ = 0;
  size_t room_ = 0;
  frontier_type_t kind_ = node_frontier;
  bool exposed_ = false;
};

}  // namespace gunrock
