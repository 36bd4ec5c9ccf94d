// mini_amd/libmgx_loopback.so -- the in-process stand-in for RCCL (include/mgx/comm_loopback.hpp) as a library of its own.
// TEST INFRASTRUCTURE: G host threads of one process as G ranks, so that the loops that enqueue push -> collective -> merge from
// C++ (bfs_dist2.hpp, sssp_dist.hpp) can run with 2 .. 64 ranks on a one-GPU box.  Until round 6 this code was compiled into the
// product library; now libmgx.so only knows how an id that names a loopback world looks (comm.hpp) and, when it meets one, takes
// the ten entry points below from this library (dlopen next to itself) -- a product build without this file refuses such ids.
#include "../../include/mgx/comm_loopback.hpp"

extern "C" __attribute__((visibility("default"))) int mgx_loopback_table(void** out, int cap) {
  namespace lb = mgx::loopback;
  void* t[10] = {(void*)lb::GetUniqueId, (void*)lb::CommInitRank, (void*)lb::CommDestroy, (void*)lb::AllGather, (void*)lb::Send,
                 (void*)lb::Recv,        (void*)lb::GroupStart,   (void*)lb::GroupEnd,    (void*)lb::GetErrorString, (void*)lb::rounds};
  for (int i = 0; i < 10 && i < cap; ++i) out[i] = t[i];
  return 10;
}
