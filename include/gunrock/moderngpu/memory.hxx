// Forwarding header: code written against mini-gunrock includes "moderngpu/memory.hxx" directly (graph.hxx:11-13,
// kcore/kcore_problem.hxx:5) for mgpu::standard_context_t / mem_t / fill / to_mem / from_mem / dtoh / transform.
// Here those names are provided by the mgx runtime (namespace mgpu = mgx), hand-written for gfx950; nothing of
// moderngpu is used or reimplemented beyond that interface.
#pragma once
#include "../../mgx/runtime.hpp"
