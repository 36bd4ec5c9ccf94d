// mgx/runtime.hpp -- host-side memory/context layer of the engine (HIP, gfx950).
//
// Re-states the slice of the reference's memory layer that its data model touches
// (SURVEY 2.3): mem_t<T>, standard_context_t, to_mem/from_mem/fill/dtoh/htod/dtod
// (reference call sites: gunrock/src/graph.hxx:40-83, frontier.hxx:18-79,
// bfs/bfs_problem.hxx:42-50, advance.hxx:30,43).  Differences by design:
//   * operators never allocate: every context owns a scratch arena and a pinned
//     host mailbox, sized when graphs/frontiers are created (the reference
//     allocates 4-5 temporaries per superstep);
//   * errors are status codes / exceptions, never exit() (frontier.hxx:53-59).
#pragma once
#include <cstdlib>
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <exception>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../mgx.h"

namespace mgx {

// Rows of at least this many entries are "long": they live a second time as unit blocks (64-entry units of one row each)
// and are streamed; shorter rows are walked vertex by vertex by degree class.  32 since round 4 (64 before): with 3-byte
// entries a half-empty unit still costs less than the short-row walk's latency -- RMAT-22, ms per traversal at 64 / 48 / 32 /
// 24 / 16: 0.3171 / 0.3185 / 0.3087 / 0.3081 / 0.3098 (MGX_BFS_LONG_MIN overrides; at most 64: a unit is 64 entries).
constexpr int LONG_MIN_DEFAULT = 32;

struct hip_error : std::runtime_error {
  hipError_t code;
  hip_error(hipError_t c, const char* what_, const char* file, int line)
      : std::runtime_error(std::string(what_) + ": " + hipGetErrorString(c) + " @" + file + ":" +
                           std::to_string(line)),
        code(c) {}
};
struct mgx_error : std::runtime_error {
  int code;
  mgx_error(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

#define MGX_HIP(expr)                                                  \
  do {                                                                 \
    hipError_t _e = (expr);                                            \
    if (_e != hipSuccess) throw ::mgx::hip_error(_e, #expr, __FILE__, __LINE__); \
  } while (0)

// A kernel launch reports a bad configuration (too much dynamic LDS, a grid the device refuses) only through
// hipGetLastError: checked after the launches of a batch, before their results are read back -- an unchecked failed
// launch leaves the control block untouched and would read as "traversal finished".
#define MGX_CHECK_LAUNCH(what)                                                          \
  do {                                                                                  \
    hipError_t _e = hipGetLastError();                                                  \
    if (_e != hipSuccess) throw ::mgx::hip_error(_e, what, __FILE__, __LINE__);         \
  } while (0)

// Per-DEVICE one-time setup (hipFuncSetAttribute is per device: a second context on another GPU of the same process
// needs it again).  `device_once_t once(seen); if (once) { ...set the attributes... }` -- true for ONE caller per device and
// tag; a second host thread that arrives while the first is still inside the block waits for it (the rank threads of a
// loopback world launch the same kernels at the same moment: "somebody has started setting the attributes" is not enough
// to launch on).  The tag is marked when the block is left without an exception.
struct device_once_t {
  unsigned char* slot = nullptr;
  bool first = false;
  std::unique_lock<std::mutex> lk;
  static std::mutex& mu() { static std::mutex m; return m; }
  explicit device_once_t(unsigned char (&seen)[64]) {
    int dev = 0;
    MGX_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { first = true; return; }
    if (__atomic_load_n(&seen[dev], __ATOMIC_ACQUIRE)) return;
    lk = std::unique_lock<std::mutex>(mu());
    if (__atomic_load_n(&seen[dev], __ATOMIC_ACQUIRE)) { lk.unlock(); return; }
    slot = &seen[dev];
    first = true;
  }
  device_once_t(const device_once_t&) = delete;
  device_once_t& operator=(const device_once_t&) = delete;
  ~device_once_t() { if (slot && !std::uncaught_exceptions()) __atomic_store_n(slot, (unsigned char)1, __ATOMIC_RELEASE); }
  explicit operator bool() const { return first; }
};

// ---------------------------------------------------------------------------
// context: one device, one stream, a scratch arena, a pinned mailbox
// ---------------------------------------------------------------------------
// one address per functor type: how an operator names "the functor I was instantiated with" in a host-side record
template <typename F>
struct functor_tag_t { static const char id; };
template <typename F>
const char functor_tag_t<F>::id = 0;

struct context_t {
  virtual ~context_t() {}
  virtual hipStream_t stream() const = 0;
};

// The stream the context-less copies below (htod / dtoh / dtod, the reference's signatures: graph.hxx:40-83,
// frontier.hxx:65-79) are ordered on: the stream of the context this host thread used last (set by
// standard_context_t's constructor / set_stream and by every C-ABI entry point).  A blocking hipMemcpy on the NULL
// stream is NOT ordered against a non-blocking stream (hipStreamNonBlocking, any torch.cuda.Stream()): a D2D copy
// could overtake the kernel that fills its source, an H2D copy could overwrite a buffer a queued kernel still reads.
inline hipStream_t& current_stream_ref() {
  static thread_local hipStream_t s = nullptr;
  return s;
}
inline hipStream_t current_stream() { return current_stream_ref(); }
inline void set_current_stream(hipStream_t s) { current_stream_ref() = s; }

// Every write into a frontier's buffer from OUTSIDE an operator (frontier_t::load, the C-ABI's load / fill / fill_iota, a device
// pointer handed to the caller) moves this counter on: what an operator remembered about a frontier's contents (the advance's
// keep-ballots for the filter behind it) is only good while it stands still.  Host-side, one per process.
inline unsigned long long& frontier_generation() {
  static unsigned long long g = 0;
  return g;
}
inline void frontier_touched() { ++frontier_generation(); }

struct standard_context_t : context_t {
  int device = 0;
  hipStream_t _stream = nullptr;   // nullptr == the legacy default stream, like the reference
  bool own_stream = false;
  // scratch arena (device) -- grows only outside operators
  void* scratch = nullptr;
  size_t scratch_bytes = 0;
  // pinned mailbox for the 8-byte count read-backs (advance.hxx:43, filter.hxx:21).  Device-visible: the kernel that
  // produces a count stores it here itself (a hipMemcpyAsync D2H of 8 bytes is a blit kernel of its own: ~20 us)
  long long* mailbox = nullptr;
  // mailbox[1]: the sequence number of the last count a kernel delivered (scan.hpp).  The host used to wait for the stream
  // (hipStreamSynchronize: ~10-20 us of runtime wake-up per operator call, two calls per superstep of an enactor) -- now it
  // spins on this word, which the producing kernel stores at system scope behind the count (mailbox_spin = false: the old wait)
  long long mailbox_seq = 0;
  // The scratch arena is shared by every operator of the context; scratch_epoch counts who wrote it.  `keep` is what an advance
  // left there for the filter behind it (gunrock/advance.hxx -> filter.hxx): the keep-ballots of its output slots, valid only for
  // that very frontier, iteration and functor, and only while nobody else has touched the arena since.
  unsigned long long scratch_epoch = 0;
  struct keep_record_t {
    const void* data = nullptr;
    long long n = 0;
    int iteration = 0;
    const void* functor = nullptr;
    unsigned long long epoch = 0;
    bool valid = false;
    unsigned long long generation = 0;     // frontier_generation() when the ballots were left
  } keep;
  bool mailbox_spin = true;
  int num_cus = 256;
  // single-pass scans (scan.hpp): one 64-bit status word per tile, tagged with the launch's epoch so that the array never
  // needs clearing, and the dynamic tile counter.  Private to those kernels (nothing else writes here: a stale word
  // can only carry an OLDER epoch).
  unsigned long long* lookback_status = nullptr;
  size_t lookback_tiles = 0;
  unsigned* lookback_ticket = nullptr;      // monotonically increasing across launches; a launch subtracts its base
  unsigned lookback_ticket_base = 0;        // (host mirror: tickets handed out by the launches enqueued so far)
  unsigned lookback_epoch = 0;
  // the neighbour-reduce's verdict word (mgx/nreduce.hpp): lives behind the ticket counter in the same 64-byte allocation,
  // holds the epoch of the last full-frontier call whose frontier was not the iota (0: none yet)
  unsigned nr_epoch = 0;
  unsigned* nr_flag() const { return lookback_ticket + 8; }
  // ... and behind that the degree sum of the subset frontiers (64 bits at byte 48; only ever added to: nr_edges_base is what it
  // held before the call in flight)
  unsigned long long* nr_edges() const { return (unsigned long long*)(lookback_ticket + 12); }
  unsigned long long nr_edges_base = 0;
  unsigned next_nr_epoch() {
    if (++nr_epoch == 0u) nr_epoch = 1u;      // (2^32 calls: a stale word could only name the call 2^32 - 1 before this one)
    return nr_epoch;
  }

  explicit standard_context_t(bool print_prop = false, hipStream_t s = nullptr) : _stream(s) {
    set_current_stream(s);
    MGX_HIP(hipGetDevice(&device));
    hipDeviceProp_t prop;
    MGX_HIP(hipGetDeviceProperties(&prop, device));
    num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if (print_prop) std::printf("%s : %d CUs\n", prop.name, num_cus);
    MGX_HIP(hipHostMalloc((void**)&mailbox, 64 * sizeof(long long), hipHostMallocDefault));
    for (int i = 0; i < 64; ++i) mailbox[i] = 0;
    reserve_scratch(1 << 20);
  }
  standard_context_t(const standard_context_t&) = delete;
  standard_context_t& operator=(const standard_context_t&) = delete;
  ~standard_context_t() override {
    if (lookback_status) (void)hipFree(lookback_status);
    if (lookback_ticket) (void)hipFree(lookback_ticket);
    if (scratch) (void)hipFree(scratch);
    if (mailbox) (void)hipHostFree(mailbox);
  }
  hipStream_t stream() const override { return _stream; }
  // Work already enqueued by this context is finished first: the scans' look-back tickets and epochs (scan.hpp), the
  // scratch arena and the mailbox assume that all launches of a context are serialised on ONE stream.
  void set_stream(hipStream_t s) {
    if (s != _stream) (void)hipStreamSynchronize(_stream);
    _stream = s;
    set_current_stream(s);
  }
  void make_current() const { set_current_stream(_stream); }
  void synchronize() { MGX_HIP(hipStreamSynchronize(_stream)); }
  // wait for the count a kernel of this context's stream delivers under sequence number `seq` (mailbox[0] then holds it)
  void mailbox_wait(long long seq) {
    if (mailbox_spin) {
      volatile long long* const flag = mailbox + 1;
      long long spins = 0;
      while (*flag != seq) {
        if (++spins > 20000000LL) { MGX_HIP(hipStreamSynchronize(_stream)); break; }      // (a failed launch: let the runtime report it)
        __builtin_ia32_pause();
      }
      __atomic_thread_fence(__ATOMIC_ACQUIRE);
    } else {
      MGX_HIP(hipStreamSynchronize(_stream));
    }
  }

  // Make sure the arena holds `bytes`.  Called from constructors of graphs,
  // frontiers and problems -- never from an operator in steady state.
  void reserve_scratch(size_t bytes) {
    if (bytes <= scratch_bytes) return;
    bytes = (bytes + 4095) & ~size_t(4095);
    // the new arena first: a failed allocation leaves the old one (and its size) in place.  Whatever an operator left in the old
    // arena for the next one (the advance's keep-ballots, `keep`) is gone with it: the epoch moves on.
    void* fresh = nullptr;
    if (scratch) MGX_HIP(hipStreamSynchronize(_stream));
    if (hipMalloc(&fresh, bytes) != hipSuccess) {
      (void)hipGetLastError();
      if (scratch) { (void)hipFree(scratch); scratch = nullptr; scratch_bytes = 0; }     // (make room and try once more)
      MGX_HIP(hipMalloc(&fresh, bytes));
    } else if (scratch) {
      MGX_HIP(hipFree(scratch));
    }
    scratch = fresh;
    scratch_bytes = bytes;
    ++scratch_epoch;
    keep.valid = false;
    // status words for a scan over as many items as this arena serves (scan_scratch_bytes: >= n / 8 bytes for n items,
    // 2048 items per tile)
    const size_t tiles = bytes / 256 + 64;
    if (tiles > lookback_tiles) {
      if (lookback_status) { MGX_HIP(hipStreamSynchronize(_stream)); MGX_HIP(hipFree(lookback_status)); lookback_status = nullptr; }
      MGX_HIP(hipMalloc((void**)&lookback_status, tiles * sizeof(unsigned long long)));
      MGX_HIP(hipMemsetAsync(lookback_status, 0, tiles * sizeof(unsigned long long), _stream));
      lookback_tiles = tiles;
      lookback_epoch = 0;
    }
    if (!lookback_ticket) {
      MGX_HIP(hipMalloc((void**)&lookback_ticket, 64));
      MGX_HIP(hipMemsetAsync(lookback_ticket, 0, 64, _stream));
      lookback_ticket_base = 0;
    }
  }
  // epoch of the next single-pass launch: 1 .. 2^30 - 1, never 0 (cleared words); at wrap-around the words are cleared
  unsigned next_lookback_epoch() {
    if (++lookback_epoch >= (1u << 30)) {
      MGX_HIP(hipMemsetAsync(lookback_status, 0, lookback_tiles * sizeof(unsigned long long), _stream));
      lookback_epoch = 1;
    }
    return lookback_epoch;
  }
};

// ---------------------------------------------------------------------------
// mem_t<T>: RAII device array (may also borrow an external device pointer)
// ---------------------------------------------------------------------------
template <typename T>
class mem_t {
  T* _ptr = nullptr;
  size_t _size = 0;
  bool _owned = true;

 public:
  mem_t() {}
  mem_t(size_t count, context_t&) : _size(count) {
    if (count) MGX_HIP(hipMalloc((void**)&_ptr, count * sizeof(T)));
  }
  // borrow: the caller keeps ownership (used by mgx_graph_wrap_device)
  static mem_t borrow(T* p, size_t count) {
    mem_t m; m._ptr = p; m._size = count; m._owned = false; return m;
  }
  // adopt: take ownership of a hipMalloc'ed array
  static mem_t adopt(T* p, size_t count) {
    mem_t m; m._ptr = p; m._size = count; m._owned = true; return m;
  }
  mem_t(const mem_t&) = delete;
  mem_t& operator=(const mem_t&) = delete;
  mem_t(mem_t&& r) noexcept { swap(r); }
  mem_t& operator=(mem_t&& r) noexcept { swap(r); return *this; }
  ~mem_t() { if (_ptr && _owned) (void)hipFree(_ptr); }
  void swap(mem_t& r) noexcept { std::swap(_ptr, r._ptr); std::swap(_size, r._size); std::swap(_owned, r._owned); }
  T* data() const { return _ptr; }
  size_t size() const { return _size; }
  bool owned() const { return _owned; }
};

// Copies are stream-ordered: issued on `s` (default: the calling thread's current context stream, see above).
// Host <-> device copies return when the data has arrived (the host buffer may be pageable and short-lived);
// device -> device copies are asynchronous on the stream like any kernel.
template <typename T>
inline hipError_t htod(T* dst, const T* src, size_t n, hipStream_t s) {
  if (!n) return hipSuccess;
  hipError_t e = hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyHostToDevice, s);
  return e != hipSuccess ? e : hipStreamSynchronize(s);
}
template <typename T>
inline hipError_t htod(T* dst, const T* src, size_t n) { return htod(dst, src, n, current_stream()); }
template <typename T>
inline hipError_t htod(T* dst, const std::vector<T>& src) { return htod(dst, src.data(), src.size()); }
template <typename T>
inline hipError_t dtoh(T* dst, const T* src, size_t n, hipStream_t s) {
  if (!n) return hipSuccess;
  hipError_t e = hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToHost, s);
  return e != hipSuccess ? e : hipStreamSynchronize(s);
}
template <typename T>
inline hipError_t dtoh(T* dst, const T* src, size_t n) { return dtoh(dst, src, n, current_stream()); }
template <typename T>
inline hipError_t dtoh(std::vector<T>& dst, const T* src, size_t n) { dst.resize(n); return dtoh(dst.data(), src, n); }
template <typename T>
inline hipError_t dtod(T* dst, const T* src, size_t n, hipStream_t s) {
  return n ? hipMemcpyAsync(dst, src, n * sizeof(T), hipMemcpyDeviceToDevice, s) : hipSuccess;
}
template <typename T>
inline hipError_t dtod(T* dst, const T* src, size_t n) { return dtod(dst, src, n, current_stream()); }

template <typename T>
inline mem_t<T> to_mem(const std::vector<T>& h, context_t& c) {
  mem_t<T> m(h.size(), c);
  MGX_HIP(htod(m.data(), h.data(), h.size(), c.stream()));
  return m;
}
template <typename T>
inline std::vector<T> from_mem(const mem_t<T>& m) {
  std::vector<T> h;
  MGX_HIP(dtoh(h, m.data(), m.size()));
  return h;
}

// fill / fill_function / transform: generic element kernels -------------------
template <typename F>
__global__ void k_transform(F f, long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) f((int)i);
}
inline int grid_for(long long n, int block = 256, int max_blocks = 4096) {
  long long g = (n + block - 1) / block;
  if (g < 1) g = 1;
  if (g > max_blocks) g = max_blocks;
  return (int)g;
}
template <typename F>
inline void transform(F f, long long n, context_t& c) {
  if (n <= 0) return;
  hipLaunchKernelGGL(k_transform<F>, dim3(grid_for(n)), dim3(256), 0, c.stream(), f, n);
}
template <typename T>
inline mem_t<T> fill(T value, size_t n, context_t& c) {
  mem_t<T> m(n, c);
  T* p = m.data();
  transform([=] __device__(int i) { p[i] = value; }, (long long)n, c);
  return m;
}
template <typename T, typename F>
inline mem_t<T> fill_function(F f, size_t n, context_t& c) {
  mem_t<T> m(n, c);
  T* p = m.data();
  transform([=] __device__(int i) { p[i] = f(i); }, (long long)n, c);
  return m;
}

// reduction operators the operator API names (advance.hxx:40, pr_enactor.hxx:53)
template <typename T> struct plus_t    { __host__ __device__ T operator()(T a, T b) const { return a + b; } };
template <typename T> struct maximum_t { __host__ __device__ T operator()(T a, T b) const { return a > b ? a : b; } };
template <typename T> struct minimum_t { __host__ __device__ T operator()(T a, T b) const { return a < b ? a : b; } };

template <typename T>
__device__ __forceinline__ T ldg(const T* p) { return *p; }   // no read-only path on CDNA

}  // namespace mgx
