// mgx/comm_loopback.hpp -- an in-process stand-in for RCCL behind rccl_api_t (comm.hpp): the same nine entry points, G host
// threads in ONE process (one "rank" each, a stream each, any device they can copy between -- the tests put them all on one GPU).
//
// Why: RCCL refuses two ranks of a communicator on one device, the pool this library is built on has one-GPU boxes, and the loops
// that enqueue push -> collective -> merge from C++ (bfs_dist2.hpp: d2_run, sssp_dist.hpp: dsssp_run) would otherwise meet their
// second rank for the first time on the eight-GPU node.  With this table behind a communicator those very loops run with 2 .. 64
// ranks under `pytest -m gpu` -- same call sequence, same buffers, same group semantics:
//   * a collective outside a group is a ROUND of its own; ncclGroupStart .. ncclGroupEnd collect sends / receives / all-gathers
//     and NOTHING moves before the outermost GroupEnd (what the slice exchange relies on);
//   * a round: the rank waits for its stream (its send buffers are final, nobody still reads its receive buffers), publishes its
//     operations and PULLS what it is to receive from the peers' buffers with device copies on its own stream, then waits for
//     the copies; a peer may touch its send buffer again only once the receiver has said so.  Sends / receives meet pairwise
//     (receive j from peer p is matched with p's j-th send to this rank; a rank without messages takes no part, as in RCCL),
//     all-gathers meet the whole world at a host barrier (all-gather k with everybody's k-th); the byte counts must agree.
//     Host-synchronous where RCCL is stream-ordered: a superset of the ordering RCCL gives, so a loop that is
//     right here can still be wrong in what it overlaps -- but not in what it sends where, in its buffer arithmetic, its group
//     structure or its termination protocol, which is what the first contact with eight ranks would otherwise test;
//   * a rank that does not show up (its thread died of an error, the loops disagree about the sequence of collectives) does not
//     hang the others: every wait has a deadline (MGX_LOOPBACK_TIMEOUT_S, 120 s), after which the world is BROKEN and every call
//     on it returns ncclSystemError.
// A loopback communicator is made from a loopback id (mgx_comm_loopback_id): the id
// carries a magic word, mgx_comm_create reads it and picks this table instead of RCCL's.  Not a transport: nothing here is meant
// to be fast, and nothing on a product path selects it by itself.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include "env.hpp"

namespace mgx {
namespace loopback {

constexpr unsigned long long LOOPBACK_MAGIC = 0x42504F4F4C58474Dull;       // "MGXLOOPB" as the id's first eight bytes: not how an RCCL id starts (comm.hpp -- never in one translation unit with this file -- knows the same word)

enum op_kind_t : int { OP_SEND = 0, OP_RECV = 1, OP_ALLGATHER = 2 };
struct op_t {
  int kind;
  const void* src;     // send / all-gather: this rank's data
  void* dst;           // recv / all-gather: where the data goes on this rank
  size_t bytes;        // send / recv: the message; all-gather: ONE rank's share
  int peer;            // send / recv
};

struct mail_t {                       // the messages of one (sender, receiver) pair, in the order they were sent
  std::deque<op_t> q;
  unsigned long long posted = 0, done = 0;
};

struct world_t {
  int nranks = 0;
  std::mutex mu;
  std::condition_variable cv;
  int joined = 0;
  std::vector<char> taken;          // rank r has joined (CommInitRank refuses a second thread with the same rank)
  int arrived = 0;
  unsigned long long generation = 0;
  bool broken = false;
  int refs = 0;
  unsigned long long key = 0;
  std::vector<std::vector<op_t>> posted;            // all-gathers: what every rank brought to the round
  std::vector<mail_t> mail;                         // point-to-point: [sender * nranks + receiver]
  std::atomic<unsigned long long> rounds{0};        // (diagnostics: rounds completed by rank 0)
};

struct lcomm_t {
  unsigned long long magic = LOOPBACK_MAGIC;
  std::shared_ptr<world_t> w;
  int rank = 0;
};

inline std::chrono::seconds timeout() {
  static const long s = [] { const char* e = mgx::env("MGX_LOOPBACK_TIMEOUT_S"); const long v = e ? std::atol(e) : 120; return v > 0 ? v : 120; }();
  return std::chrono::seconds(s);
}

struct registry_t {
  std::mutex mu;
  std::map<unsigned long long, std::shared_ptr<world_t>> worlds;
  std::atomic<unsigned long long> next{1};
  static registry_t& get() { static registry_t r; return r; }
};

// all ranks of the world, or nobody: false once the world is broken (a rank missed the deadline)
inline bool barrier(world_t& w) {
  std::unique_lock<std::mutex> lk(w.mu);
  if (w.broken) return false;
  const unsigned long long gen = w.generation;
  if (++w.arrived == w.nranks) {
    w.arrived = 0;
    ++w.generation;
    w.cv.notify_all();
    return true;
  }
  if (!w.cv.wait_for(lk, timeout(), [&] { return w.generation != gen || w.broken; })) {
    w.broken = true;
    w.cv.notify_all();
    return false;
  }
  return w.generation != gen;
}

inline size_t type_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

// the calling thread's open group: operations wait here for the outermost GroupEnd
struct group_state_t {
  int depth = 0;
  lcomm_t* comm = nullptr;
  hipStream_t stream = nullptr;
  std::vector<op_t> ops;
  bool failed = false;
};
inline group_state_t& group_state() { static thread_local group_state_t g; return g; }

// One round = what a rank handed over between the outermost GroupStart and GroupEnd (or one call outside a group).
// Sends and receives are POINT-TO-POINT, as in RCCL: only the two ranks of a message meet (a rank with nothing to send or to
// receive in a superstep issues no group at all -- sssp_dist.hpp -- and must not be waited for); all-gathers are collectives of
// the whole world.
//   p2p: the rank posts its sends into the mailbox of each (sender, receiver) pair -- nothing blocks yet --, then takes its
//        receives in order (receive j from peer p = p's j-th unconsumed send to this rank; waits for it to be posted; sizes must
//        agree), copies on its own stream, waits for the stream, acknowledges, and finally waits until every send of its own has
//        been acknowledged: only then may it touch its send buffers again.  Posting before receiving is what makes a group of
//        mutual sends deadlock-free, as RCCL's groups are.
//   all-gather k of the round is matched with every rank's k-th all-gather of ITS round: two barriers around the copies.
inline ncclResult_t run_round(lcomm_t* c, hipStream_t stream, const std::vector<op_t>& ops) {
  world_t& w = *c->w;
  const int me = c->rank, R = w.nranks;
  if (hipStreamSynchronize(stream) != hipSuccess) return ncclUnhandledCudaError;
  auto fail = [&](ncclResult_t r) {
    std::lock_guard<std::mutex> g(w.mu);
    w.broken = true;
    w.cv.notify_all();
    return r;
  };
  bool any_p2p = false, any_ag = false;
  for (const op_t& op : ops) { any_p2p |= op.kind != OP_ALLGATHER; any_ag |= op.kind == OP_ALLGATHER; }
  if (any_p2p) {
    std::vector<unsigned long long> wait_done((size_t)R, 0);        // per receiver: acknowledgements this rank waits for
    std::vector<int> sent_to((size_t)R, 0);
    {
      std::lock_guard<std::mutex> g(w.mu);
      if (w.broken) return ncclSystemError;
      for (const op_t& op : ops)
        if (op.kind == OP_SEND) {
          mail_t& m = w.mail[(size_t)me * R + op.peer];
          m.q.push_back(op);
          wait_done[(size_t)op.peer] = ++m.posted;
          sent_to[(size_t)op.peer] = 1;
        }
      w.cv.notify_all();
    }
    std::vector<int> to_ack((size_t)R, 0);
    for (const op_t& op : ops) {
      if (op.kind != OP_RECV) continue;
      op_t from;
      {
        std::unique_lock<std::mutex> lk(w.mu);
        mail_t& m = w.mail[(size_t)op.peer * R + me];
        if (!w.cv.wait_for(lk, timeout(), [&] { return !m.q.empty() || w.broken; })) { w.broken = true; w.cv.notify_all(); return ncclSystemError; }
        if (w.broken) return ncclSystemError;
        from = m.q.front();
        m.q.pop_front();
      }
      if (from.bytes != op.bytes) return fail(ncclInvalidUsage);
      if (op.bytes && hipMemcpyAsync(op.dst, from.src, op.bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) return fail(ncclUnhandledCudaError);
      ++to_ack[(size_t)op.peer];
    }
    if (hipStreamSynchronize(stream) != hipSuccess) return fail(ncclUnhandledCudaError);
    {
      std::unique_lock<std::mutex> lk(w.mu);
      for (int p = 0; p < R; ++p)
        if (to_ack[(size_t)p]) w.mail[(size_t)p * R + me].done += (unsigned long long)to_ack[(size_t)p];
      w.cv.notify_all();
      for (int p = 0; p < R; ++p) {
        if (!sent_to[(size_t)p]) continue;
        mail_t& m = w.mail[(size_t)me * R + p];
        if (!w.cv.wait_for(lk, timeout(), [&] { return m.done >= wait_done[(size_t)p] || w.broken; })) { w.broken = true; w.cv.notify_all(); return ncclSystemError; }
        if (w.broken) return ncclSystemError;
      }
    }
  }
  if (any_ag) {
    {
      std::lock_guard<std::mutex> g(w.mu);
      if (w.broken) return ncclSystemError;
      w.posted[(size_t)me] = ops;
    }
    if (!barrier(w)) return ncclSystemError;
    // (the peers' lists are final and stay so until the second barrier)
    bool ok = true;
    int ag_seen = 0;
    for (const op_t& op : ops) {
      if (op.kind != OP_ALLGATHER) continue;
      const int want = ag_seen++;
      for (int p = 0; p < R && ok; ++p) {
        const op_t* match = nullptr;
        int k = 0;
        for (const op_t& q : w.posted[(size_t)p])
          if (q.kind == OP_ALLGATHER && k++ == want) { match = &q; break; }
        if (!match || match->bytes != op.bytes) { ok = false; break; }
        char* const to = (char*)op.dst + (size_t)p * op.bytes;
        if (op.bytes && (const void*)to != match->src &&            // (in place: a rank's share already lies where it belongs)
            hipMemcpyAsync(to, match->src, op.bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess) ok = false;
      }
      if (!ok) break;
    }
    if (hipStreamSynchronize(stream) != hipSuccess) ok = false;
    if (!ok) return fail(ncclInvalidUsage);
    if (!barrier(w)) return ncclSystemError;
  }
  if (me == 0) w.rounds.fetch_add(1, std::memory_order_relaxed);
  return ncclSuccess;
}

inline ncclResult_t submit(ncclComm_t comm, hipStream_t stream, const op_t& op) {
  lcomm_t* c = (lcomm_t*)comm;
  if (!c || c->magic != LOOPBACK_MAGIC || !c->w) return ncclInvalidArgument;
  if ((op.kind == OP_SEND || op.kind == OP_RECV) && (op.peer < 0 || op.peer >= c->w->nranks)) return ncclInvalidArgument;
  group_state_t& g = group_state();
  if (g.depth > 0) {
    if (g.comm && (g.comm != c || g.stream != stream)) { g.failed = true; return ncclInvalidUsage; }     // (one communicator and stream per group)
    g.comm = c; g.stream = stream;
    g.ops.push_back(op);
    return ncclSuccess;
  }
  return run_round(c, stream, std::vector<op_t>{op});
}

inline ncclResult_t GetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  std::memset(id->internal, 0, NCCL_UNIQUE_ID_BYTES);
  const unsigned long long magic = LOOPBACK_MAGIC;
  const unsigned long long key = registry_t::get().next.fetch_add(1) | ((unsigned long long)(std::chrono::steady_clock::now().time_since_epoch().count() & 0xFFFFFFll) << 32);
  std::memcpy(id->internal, &magic, 8);
  std::memcpy(id->internal + 8, &key, 8);
  return ncclSuccess;
}
inline bool is_loopback_id(const unsigned char* id128) {
  unsigned long long magic = 0;
  std::memcpy(&magic, id128, 8);
  return magic == LOOPBACK_MAGIC;
}

// collective: returns when all `nranks` threads have joined the world the id names (or the deadline has passed)
inline ncclResult_t CommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (!out || nranks < 1 || rank < 0 || rank >= nranks || !is_loopback_id((const unsigned char*)id.internal)) return ncclInvalidArgument;
  unsigned long long key = 0;
  std::memcpy(&key, id.internal + 8, 8);
  std::shared_ptr<world_t> w;
  {
    registry_t& reg = registry_t::get();
    std::lock_guard<std::mutex> g(reg.mu);
    auto it = reg.worlds.find(key);
    if (it == reg.worlds.end()) {
      w = std::make_shared<world_t>();
      w->nranks = nranks; w->key = key;
      w->posted.resize((size_t)nranks);
      w->mail.resize((size_t)nranks * (size_t)nranks);
      reg.worlds[key] = w;
    } else w = it->second;
  }
  bool failed = false, last = false;
  {
    std::unique_lock<std::mutex> lk(w->mu);
    if (w->nranks != nranks || w->broken || w->joined >= nranks) return ncclInvalidArgument;
    if (w->taken.size() != (size_t)nranks) w->taken.assign((size_t)nranks, 0);
    if (w->taken[(size_t)rank]) return ncclInvalidArgument;          // (two threads with one rank: their mailboxes would alias silently)
    w->taken[(size_t)rank] = 1;
    ++w->joined; ++w->refs;
    if (w->joined == nranks) w->cv.notify_all();
    else if (!w->cv.wait_for(lk, timeout(), [&] { return w->joined == w->nranks || w->broken; })) { w->broken = true; w->cv.notify_all(); }
    if (w->broken) { failed = true; last = --w->refs == 0; }
  }
  if (failed) {
    if (last) {                                                       // (a broken world leaves the registry with its last member)
      registry_t& reg = registry_t::get();
      std::lock_guard<std::mutex> g(reg.mu);
      reg.worlds.erase(key);
    }
    return ncclSystemError;
  }
  lcomm_t* c = new lcomm_t();
  c->w = w; c->rank = rank;
  *out = (ncclComm_t)c;
  return ncclSuccess;
}
inline ncclResult_t CommDestroy(ncclComm_t comm) {
  lcomm_t* c = (lcomm_t*)comm;
  if (!c || c->magic != LOOPBACK_MAGIC) return ncclInvalidArgument;
  {                                    // (a group this thread left open on the communicator: its operations must not outlive it)
    group_state_t& g = group_state();
    if (g.comm == c) { g.comm = nullptr; g.ops.clear(); g.failed = true; }
  }
  bool last = false;
  unsigned long long key = 0;
  if (c->w) {
    std::lock_guard<std::mutex> g(c->w->mu);
    last = --c->w->refs == 0;
    key = c->w->key;
  }
  if (last) {
    registry_t& reg = registry_t::get();
    std::lock_guard<std::mutex> g(reg.mu);
    reg.worlds.erase(key);
  }
  c->magic = 0;
  delete c;
  return ncclSuccess;
}
inline ncclResult_t AllGather(const void* send, void* recv, size_t count, ncclDataType_t t, ncclComm_t comm, hipStream_t s) {
  const size_t tb = type_bytes(t);
  if (!tb || (count && (!send || !recv))) return ncclInvalidArgument;
  return submit(comm, s, op_t{OP_ALLGATHER, send, recv, count * tb, -1});
}
inline ncclResult_t Send(const void* send, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
  const size_t tb = type_bytes(t);
  if (!tb || (count && !send)) return ncclInvalidArgument;
  return submit(comm, s, op_t{OP_SEND, send, nullptr, count * tb, peer});
}
inline ncclResult_t Recv(void* recv, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s) {
  const size_t tb = type_bytes(t);
  if (!tb || (count && !recv)) return ncclInvalidArgument;
  return submit(comm, s, op_t{OP_RECV, nullptr, recv, count * tb, peer});
}
inline ncclResult_t GroupStart() {
  group_state_t& g = group_state();
  if (g.depth++ == 0) { g.comm = nullptr; g.stream = nullptr; g.ops.clear(); g.failed = false; }
  return ncclSuccess;
}
inline ncclResult_t GroupEnd() {
  group_state_t& g = group_state();
  if (g.depth <= 0) return ncclInvalidUsage;
  if (--g.depth > 0) return ncclSuccess;
  ncclResult_t r = ncclSuccess;
  if (g.failed) r = ncclInvalidUsage;
  else if (g.comm) r = run_round(g.comm, g.stream, g.ops);
  g.comm = nullptr; g.ops.clear(); g.failed = false;
  return r;
}
inline const char* GetErrorString(ncclResult_t r) {
  switch (r) {
    case ncclSuccess: return "loopback: success";
    case ncclSystemError: return "loopback: a rank did not arrive in time or the world is broken (MGX_LOOPBACK_TIMEOUT_S)";
    case ncclInvalidUsage: return "loopback: the ranks' operations of a round do not match (counts, sizes, send without receive)";
    case ncclInvalidArgument: return "loopback: invalid argument";
    case ncclUnhandledCudaError: return "loopback: HIP error";
    default: return "loopback: error";
  }
}
// rounds the world of this communicator has completed (tests: how many collectives did a traversal take?)
inline unsigned long long rounds(ncclComm_t comm) {
  lcomm_t* c = (lcomm_t*)comm;
  return (c && c->magic == LOOPBACK_MAGIC && c->w) ? c->w->rounds.load(std::memory_order_relaxed) : 0ull;
}

}  // namespace loopback
}  // namespace mgx
