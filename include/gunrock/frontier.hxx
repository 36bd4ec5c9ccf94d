// gunrock/frontier.hxx -- frontier_t<T>: a device id buffer with size / capacity.
// Drop-in for the reference's gunrock/src/frontier.hxx:12-99 (same members: load(mem_t&),
// load(vector), resize, capacity, size, type, data, swap).  Differences: capacity overflow
// throws mgx::mgx_error(MGX_E_FRONTIER_OVERFLOW) where the reference printf()s and exit(0)s
// (:53-59, :84-89), and constructing a frontier reserves the context's scratch arena for a
// scan/compaction of `capacity` items so operators never allocate.
#pragma once

#include "graph.hxx"
#include "../mgx/scan.hpp"
#include "../mgx/lbs.hpp"

namespace gunrock {

enum frontier_type_t { edge_frontier = 0, node_frontier = 1 };

template <typename type_t>
class frontier_t {
  size_t _size;
  size_t _capacity;
  frontier_type_t _type;
  std::shared_ptr<mem_t<type_t>> _data;

  static void overflow(const char* what, size_t cap, size_t want) {
    throw mgx::mgx_error(MGX_E_FRONTIER_OVERFLOW, std::string("Overflow during frontier ") + what + ". Capacity is " +
                                                      std::to_string(cap) + ", size of the data is " +
                                                      std::to_string(want) + ".");
  }

 public:
  void swap(frontier_t& rhs) {
    std::swap(_size, rhs._size);
    std::swap(_capacity, rhs._capacity);
    std::swap(_type, rhs._type);
    _data.swap(rhs._data);
  }

  frontier_t() : _size(0), _capacity(0), _type(node_frontier), _data(std::make_shared<mem_t<type_t>>()) {}
  frontier_t& operator=(const frontier_t& rhs) = delete;
  frontier_t(const frontier_t& rhs) = delete;

  frontier_t(context_t& context, size_t capacity, size_t size = 0, frontier_type_t type = node_frontier)
      : _size(size), _capacity(capacity), _type(type) {
    _data.reset(new mem_t<type_t>(capacity, context));
    if (auto* sc = dynamic_cast<standard_context_t*>(&context))
      sc->reserve_scratch(mgx::scan_scratch_bytes((long long)capacity) +
                          mgx::segreduce_scratch_bytes((long long)capacity, 8));
  }

  frontier_t(frontier_t&& rhs) : frontier_t() { swap(rhs); }
  frontier_t& operator=(frontier_t&& rhs) {
    swap(rhs);
    return *this;
  }
  ~frontier_t() {}

  hipError_t load(mem_t<type_t>& target) {
    if (target.size() > _data->size()) overflow("loading", _data->size(), target.size());
    hipError_t result = dtod(_data->data(), target.data(), target.size());
    _size = target.size();
    return result;
  }
  hipError_t load(const std::vector<type_t>& target) {
    if (target.size() > _data->size()) overflow("loading", _data->size(), target.size());
    hipError_t result = htod(_data->data(), target);
    _size = target.size();
    return result;
  }
  void resize(size_t size) {
    if (size > _capacity) overflow("resizing", _capacity, size);
    _size = size;
  }
  size_t capacity() const { return _capacity; }
  size_t size() const { return _size; }
  frontier_type_t type() const { return _type; }
  std::shared_ptr<mem_t<type_t>> data() const { return _data; }
};

}  // namespace gunrock
