// Host stand-in for the ONE purpose of tools/regen_goldens.sh: letting the reference's own load_graph / bfs_problem_t::cpu /
// sssp_problem_t::cpu (which sit in headers that include moderngpu, absent from /root/reference) compile with g++ in
// the build container, so that their OUTPUTS can be recorded as golden vectors.  Own code, not
// moderngpu's; never part of the product, the oracle or oracle/_ref; nothing here travels to the GPU box.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <limits>
#include <memory>
#include <vector>
