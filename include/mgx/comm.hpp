// mgx/comm.hpp -- RCCL from C++, without a link-time dependency.
//
// The partitioned traversal's per-level loop (bfs_dist2.hpp: push -> exchange -> merge) used to be driven from Python:
// a torch.distributed collective between two ctypes calls per level, ~0.1 ms of interpreter per level.  With a
// communicator of its own the library enqueues push, the RCCL collective and the merge of a whole BATCH of levels on
// the context's stream back to back and synchronises once per batch.
// RCCL is taken from the process: PyTorch ships its own librccl.so.1 and has loaded it by the time a process group
// exists; dlopen finds that copy by soname (a second copy of the runtime in one process would see no peers), and falls
// back to the system's /opt/rocm/lib/librccl.so.1 for hosts without torch.  Only the handful of entry points below.
#pragma once
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <string>

#include "runtime.hpp"

namespace mgx {

// An id whose first eight bytes are this word names a world of the in-process stand-in for RCCL (comm_loopback.hpp: test
// infrastructure, mini_amd/libmgx_loopback.so) -- not how an RCCL id starts.
namespace loopback {
constexpr unsigned long long LOOPBACK_MAGIC = 0x42504F4F4C58474Dull;       // "MGXLOOPB"
inline bool is_loopback_id(const unsigned char* id128) {
  unsigned long long magic = 0;
  __builtin_memcpy(&magic, id128, 8);
  return magic == LOOPBACK_MAGIC;
}
}  // namespace loopback

struct rccl_api_t {
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  void* handle = nullptr;
  std::string where;

  static rccl_api_t& get() {
    static rccl_api_t api;
    static std::once_flag once;
    std::call_once(once, [] {
      const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
      for (int pass = 0; pass < 2 && !api.handle; ++pass)
        for (const char* nm : names) {
          // pass 0: only a copy that is already in the process (torch's); pass 1: load one
          api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
          if (api.handle) { api.where = std::string(nm) + (pass == 0 ? " (already loaded)" : ""); break; }
        }
      if (!api.handle) return;
#define MGX_RCCL_SYM(field, sym) api.field = (decltype(api.field))dlsym(api.handle, sym)
      MGX_RCCL_SYM(GetUniqueId, "ncclGetUniqueId");
      MGX_RCCL_SYM(CommInitRank, "ncclCommInitRank");
      MGX_RCCL_SYM(CommDestroy, "ncclCommDestroy");
      MGX_RCCL_SYM(AllGather, "ncclAllGather");
      MGX_RCCL_SYM(Send, "ncclSend");
      MGX_RCCL_SYM(Recv, "ncclRecv");
      MGX_RCCL_SYM(GroupStart, "ncclGroupStart");
      MGX_RCCL_SYM(GroupEnd, "ncclGroupEnd");
      MGX_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef MGX_RCCL_SYM
    });
    return api;
  }
  bool ok() const {
    return handle && GetUniqueId && CommInitRank && CommDestroy && AllGather && Send && Recv && GroupStart && GroupEnd;
  }
  // The in-process stand-in (comm_loopback.hpp): the same nine pointers, G host threads for G ranks.  Only a communicator made
  // from a loopback id carries this table (comm_t::api); nothing selects it by default.
  // (round 6: the stand-in is a library of its own, mini_amd/libmgx_loopback.so -- the product carries none of it.  It is looked for
  //  next to the library this code is in; a build without it has an empty table here: ok() is false, a loopback id is refused.)
  unsigned long long (*rounds)(ncclComm_t) = nullptr;     // loopback only: collective rounds the communicator's world has completed
  static const rccl_api_t& loopback() {
    static const rccl_api_t api = [] {
      rccl_api_t a;
      Dl_info self;
      std::string dir;
      if (dladdr((const void*)&rccl_api_t::loopback, &self) && self.dli_fname) {
        dir = self.dli_fname;
        const size_t slash = dir.rfind('/');
        dir = slash == std::string::npos ? std::string() : dir.substr(0, slash + 1);
      }
      const std::string path = dir + "libmgx_loopback.so";
      void* h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (!h) return a;
      typedef int (*table_fn)(void**, int);
      table_fn fn = (table_fn)dlsym(h, "mgx_loopback_table");
      void* t[10] = {};
      if (!fn || fn(t, 10) < 10) return a;
      a.GetUniqueId = (decltype(a.GetUniqueId))t[0];
      a.CommInitRank = (decltype(a.CommInitRank))t[1];
      a.CommDestroy = (decltype(a.CommDestroy))t[2];
      a.AllGather = (decltype(a.AllGather))t[3];
      a.Send = (decltype(a.Send))t[4];
      a.Recv = (decltype(a.Recv))t[5];
      a.GroupStart = (decltype(a.GroupStart))t[6];
      a.GroupEnd = (decltype(a.GroupEnd))t[7];
      a.GetErrorString = (decltype(a.GetErrorString))t[8];
      a.rounds = (decltype(a.rounds))t[9];
      a.handle = h;
      a.where = "loopback (in-process stand-in: " + path + ")";
      return a;
    }();
    return api;
  }
};

// (expects `api`, the table the call went through, in scope)
#define MGX_RCCL(expr)                                                                                         \
  do {                                                                                                         \
    ncclResult_t _r = (expr);                                                                                  \
    if (_r != ncclSuccess)                                                                                     \
      throw ::mgx::mgx_error(MGX_E_HIP, std::string(#expr) + ": " + (api.GetErrorString ? api.GetErrorString(_r) : "RCCL error")); \
  } while (0)

// ncclGroupStart ... ncclGroupEnd as a scope: end() closes the group and reports its status; if the scope is left by an
// exception before that, the destructor still closes the group (its status is dropped: the first error is on its way up)
struct rccl_group_t {
  const rccl_api_t& api;
  bool open = false;
  explicit rccl_group_t(const rccl_api_t& a) : api(a) { MGX_RCCL(api.GroupStart()); open = true; }
  rccl_group_t(const rccl_group_t&) = delete;
  rccl_group_t& operator=(const rccl_group_t&) = delete;
  void end() { open = false; MGX_RCCL(api.GroupEnd()); }
  ~rccl_group_t() { if (open) (void)api.GroupEnd(); }
};

struct comm_t {
  ncclComm_t comm = nullptr;
  int ranks = 1, rank = 0;
  const rccl_api_t* api = nullptr;        // the table this communicator was made through (RCCL's or the loopback's); nullptr: RCCL's
  comm_t() {}
  comm_t(const comm_t&) = delete;
  comm_t& operator=(const comm_t&) = delete;
  const rccl_api_t& table() const { return api ? *api : rccl_api_t::get(); }
  ~comm_t() { if (comm) (void)table().CommDestroy(comm); }
};

}  // namespace mgx
