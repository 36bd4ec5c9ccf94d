// mgx/src_shapes.hpp -- what a traversal from a given source starts with: the source's degree and the shape of the level behind it
// (its distinct neighbours other than itself that have entries: their entries in all, how many of them are short and long rows).
// The fused BFS enqueues a traversal's launches before its level structure is known; these four numbers let the host pick the
// launch sequence per source (bfs_fused_run.hpp: bfs_classify_source).
//
// Until round 5 the layout builder computed them for EVERY vertex -- one wave per vertex, a kernel that read 9.8 GB on RMAT-22 --
// and kept 16 bytes per vertex in host memory (67 MB).  Now they are computed for the sources that are asked for: one launch
// (a wave per source) and one wait per batch of sources that the cache has not seen, four words per source kept in a host map.
#pragma once
#include <array>
#include <unordered_map>
#include <vector>

#include "runtime.hpp"
#include "wave.hpp"

namespace mgx {

// out[i] = (degree, level-1 entries -- 0xFFFFFFFF when they do not fit --, level-1 short rows, level-1 long rows) of ids[i]; a neighbour
// counts once (rows sorted by neighbour: a duplicate sits next to its twin), the vertex itself and neighbours without entries do not
__global__ __launch_bounds__(256) void k_src_shapes_some(const int* __restrict__ ro, const int* __restrict__ ci, int long_min,
                                                         const int* __restrict__ ids, int count, uint4* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long long wave0 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
  for (long long i = wave0; i < count; i += nwaves) {
    const int v = ids[i];
    const int r0 = ro[v], r1 = ro[v + 1];
    unsigned long long edges = 0;
    unsigned rs = 0, rl = 0;
    for (int e = r0 + lane; e < r1; e += 64) {
      const int u = ci[e];
      if (u == v || (e > r0 && ci[e - 1] == u)) continue;
      const unsigned d = (unsigned)(ro[u + 1] - ro[u]);
      if (d == 0u) continue;
      edges += d;
      if (long_min > 0 && d >= (unsigned)long_min) ++rl; else ++rs;
    }
    edges = wave_sum(edges);
    rs = wave_sum(rs);
    rl = wave_sum(rl);
    if (lane == 0) out[i] = make_uint4((unsigned)(r1 - r0), edges > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)edges, rs, rl);
  }
}

// the same for ONE source given by value (no id list to upload)
__global__ __launch_bounds__(64) void k_src_shapes_one(const int* __restrict__ ro, const int* __restrict__ ci, int long_min, int v, uint4* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int r0 = ro[v], r1 = ro[v + 1];
  unsigned long long edges = 0;
  unsigned rs = 0, rl = 0;
  for (int e = r0 + lane; e < r1; e += 64) {
    const int u = ci[e];
    if (u == v || (e > r0 && ci[e - 1] == u)) continue;
    const unsigned d = (unsigned)(ro[u + 1] - ro[u]);
    if (d == 0u) continue;
    edges += d;
    if (long_min > 0 && d >= (unsigned)long_min) ++rl; else ++rs;
  }
  edges = wave_sum(edges);
  rs = wave_sum(rs);
  rl = wave_sum(rl);
  if (lane == 0) out[0] = make_uint4((unsigned)(r1 - r0), edges > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)edges, rs, rl);
}

struct src_shape_cache_t {
  std::unordered_map<int, std::array<unsigned, 4>> seen;
  int long_min = -1;                  // the long-row threshold the cached shapes were split by
  mem_t<int> d_ids;
  mem_t<unsigned> d_out;
  size_t cap = 0;
  long long launches = 0;             // (statistics: how often the cache had to ask the device)

  void clear() { seen.clear(); long_min = -1; pending = -1; }

  // ONE source, without a wait of its own (a single-source call: mgx_bfs_run): a source the cache knows fills table[0 .. 3] and
  // returns true; one it does not know is asked for on the context's stream -- kernel + copy into pinned memory, in front of the
  // traversal's launches -- and returns false: THIS traversal gets the graph-wide launch sequence, and collect() behind its final
  // wait (the copy is older than everything the traversal enqueued) puts the shape into the cache for the next call with that source.
  // (A wait here cost a one-call-per-source loop over fresh sources 35 us per call: 0.308 -> 0.344 ms on RMAT-22.)
  unsigned* pinned = nullptr;          // 4 words, hipHostMalloc
  int pending = -1;                    // the source whose shape is on its way
  ~src_shape_cache_t() { if (pinned) (void)hipHostFree(pinned); }
  src_shape_cache_t() {}
  src_shape_cache_t(const src_shape_cache_t&) = delete;
  src_shape_cache_t& operator=(const src_shape_cache_t&) = delete;
  bool lookup_or_request(const int* ro, const int* ci, int n, int lm, int src, std::vector<unsigned>& table, standard_context_t& ctx) {
    if (lm != long_min) { seen.clear(); long_min = lm; pending = -1; }
    table.assign(4, 0u);
    table[1] = 0xFFFFFFFFu;
    if (src < 0 || src >= n) return true;                         // (classifies as "unknown")
    auto it = seen.find(src);
    if (it != seen.end()) { for (int q = 0; q < 4; ++q) table[(size_t)q] = it->second[(size_t)q]; return true; }
    if (!pinned) MGX_HIP(hipHostMalloc((void**)&pinned, 64, hipHostMallocDefault));
    // (the kernel stores straight into the pinned words: no copy of its own on the stream)
    hipLaunchKernelGGL(k_src_shapes_one, dim3(1), dim3(64), 0, ctx.stream(), ro, ci, lm, src, (uint4*)pinned);
    pending = src;
    ++launches;
    return false;
  }
  // behind a wait for the stream's work enqueued AFTER lookup_or_request: the requested shape has arrived
  void collect() {
    if (pending < 0 || !pinned) return;
    if (seen.size() > (1u << 20)) seen.clear();
    seen[pending] = {pinned[0], pinned[1], pinned[2], pinned[3]};
    pending = -1;
  }

  // table[4 i ..] <- the shape of srcs[i] (ORIGINAL ids; a source outside [0, n): degree 0, which classifies as "unknown").
  // ro / ci: the graph's CSR as loaded (device).  At most ONE launch and one wait, for the sources not seen before.
  void resolve(const int* ro, const int* ci, int n, int lm, const int* srcs, int count, std::vector<unsigned>& table, standard_context_t& ctx) {
    if (lm != long_min) { seen.clear(); long_min = lm; }
    if (seen.size() > (1u << 20)) seen.clear();                 // (a caller that walks through millions of sources: start over)
    std::vector<int> missing;
    for (int i = 0; i < count; ++i) {
      const int s = srcs[i];
      if (s < 0 || s >= n || seen.count(s)) continue;
      seen[s] = {0u, 0xFFFFFFFFu, 0u, 0u};                      // (placeholder: a duplicate in the batch is listed once)
      missing.push_back(s);
    }
    if (!missing.empty()) {
      const size_t k = missing.size();
      if (k > cap) {
        ctx.synchronize();
        cap = k < 64 ? 64 : k;
        d_ids = mem_t<int>(cap, ctx);
        d_out = mem_t<unsigned>(cap * 4, ctx);
      }
      hipStream_t s = ctx.stream();
      MGX_HIP(hipMemcpyAsync(d_ids.data(), missing.data(), k * sizeof(int), hipMemcpyHostToDevice, s));
      const unsigned grid = (unsigned)((k * 64 + 255) / 256);
      hipLaunchKernelGGL(k_src_shapes_some, dim3(grid < 4096u ? grid : 4096u), dim3(256), 0, s, ro, ci, lm, (const int*)d_ids.data(), (int)k,
                         (uint4*)d_out.data());
      std::vector<unsigned> h(k * 4);
      MGX_HIP(hipMemcpyAsync(h.data(), d_out.data(), k * 4 * sizeof(unsigned), hipMemcpyDeviceToHost, s));
      MGX_HIP(hipStreamSynchronize(s));
      for (size_t j = 0; j < k; ++j) seen[missing[j]] = {h[4 * j], h[4 * j + 1], h[4 * j + 2], h[4 * j + 3]};
      ++launches;
    }
    table.assign((size_t)count * 4, 0u);
    for (int i = 0; i < count; ++i) {
      const int s = srcs[i];
      if (s < 0 || s >= n) { table[4 * (size_t)i + 1] = 0xFFFFFFFFu; continue; }
      const std::array<unsigned, 4>& a = seen[s];
      for (int q = 0; q < 4; ++q) table[4 * (size_t)i + q] = a[q];
    }
  }
};

}  // namespace mgx
