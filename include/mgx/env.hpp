// mgx/env.hpp -- the ONE place the library reads its environment.
//
// Every switch the product (and the lab build) knows is a row of the table below: name + what it does.  `mgx::env(name)` is the only
// caller of getenv in the tree: it returns the value, NULL when the variable is unset OR EMPTY (an exported `MGX_X=` is not a zero), and
// refuses a name that is not in the table (a typo in the sources aborts at the first call instead of silently reading nothing).
// `mgx_env_switches` (include/mgx.h) hands the table to tools and tests: tests/test_capi_boundary.py checks that every MGX_* variable the
// tests, the tools and the bench scripts set is a switch the library really reads (a typo there used to test nothing, silently).
//
// The switches select between PRODUCT paths that the default thresholds pick by size -- the parity tests force each of them on small
// graphs; none changes a result.  DESIGN.md 6.1 says which families exist; the rows are the documentation of the individual ones.
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace mgx {

struct env_switch_t { const char* name; const char* what; };

inline const env_switch_t* env_switches(int* count) {
  static const env_switch_t table[] = {
    // ---- fused BFS (bfs_run_opts_t::from_env: read once per handle) and the layout it runs on
    {"MGX_BFS_DENSE", "0 / N: unit-block body for long rows off / a level takes it from frontier edges >= units / N"},
    {"MGX_BFS_VSHORT", "0 / N: short rows vertex by vertex off / threshold divisor"},
    {"MGX_BFS_LAZY", "0 / N: lazy queues off / the build writes no queues behind a push that stored >= n / N marks"},
    {"MGX_BFS_COLD", "0 / 2 (lab: 1): cold-edge pass off / on"},
    {"MGX_BFS_COLD_LISTS", "0 / 1 (lab: 2): whether the layout builds the long rows' cold-edge lists"},
    {"MGX_BFS_COLD_PACK", "0: keep the 8-byte cold pairs instead of the packed 4-byte ones"},
    {"MGX_BFS_COLD_TEST", "0 / 1: bitmap probe of cold neighbours off / on (default: by graph size)"},
    {"MGX_BFS_FLAT_LISTS", "0: a flat graph keeps the ordinary layout instead of all-entry pair lists"},
    {"MGX_BFS_HOT_UNITS", "0: no second copy of the unit blocks without the cold lists' entries"},
    {"MGX_BFS_PACK24", "0: 32-bit unit-block entries instead of the 24-bit copy"},
    {"MGX_BFS_MINI", "0 / 2: no M launches / M launches on every graph (default: from 2^22 vertices on)"},
    {"MGX_BFS_SEED_CHAIN", "0: the chain of small levels at the start runs inside slot 0's push launch"},
    {"MGX_BFS_TAIL_CHAIN", "0: no in-place chain launch behind the last slots"},
    {"MGX_BFS_TAIL_FRONT", "0: no chain in front of the last slots"},
    {"MGX_BFS_CHAIN_BIG_EDGES", "largest level an in-place chain launch runs (edges)"},
    {"MGX_BFS_CHAIN_MAX_EDGES", "largest level block 0 runs inside the push launch (0: no chains)"},
    {"MGX_BFS_MERGED_PUSH", "0: the parts of a slot's push as separate launches"},
    {"MGX_BFS_MERGED_PULL", "0: direction-optimising runs: the bottom-up sweep as a launch of its own"},
    {"MGX_BFS_DO_CHAIN", "0: direction-optimising runs: no chains of small top-down levels"},
    {"MGX_BFS_DEFER", "0 / N: deferred hot marks off / a workgroup flushes its bitmap above N claims"},
    {"MGX_BFS_DEFER_REACH", "\"mul/div\": a level defers its hot marks while reached * mul < range * div"},
    {"MGX_BFS_DEFER_WORDS", "words of the bitmap prefix whose marks are deferred (default: all)"},
    {"MGX_BFS_BUILD_LIST", "1: the list-based queue build of round 1"},
    {"MGX_BFS_SRC_PLAN", "0: no per-source launch plan"},
    {"MGX_BFS_LONG_MIN", "long-row threshold of the layout (entries)"},
    {"MGX_BFS_HOT_MIN_EDGES", "smallest level that copies the bitmap prefix to LDS (edges)"},
    {"MGX_BFS_FLAGS", "(lab) instrumented stream kernel; results wrong by design"},
    {"MGX_BFS_BUILD_DIAG", "(lab) parts of the queue build switched off; results wrong by design"},
    {"MGX_BFS_DENSE_DIAG", "(lab) parts of the unit-block body switched off; results wrong by design"},
    // ---- fused SSSP
    {"MGX_SSSP_DENSE", "0 / N: no sweep for heavy iterations / an iteration is heavy from m / N frontier edges"},
    {"MGX_SSSP_HOT_MIN_EDGES", "smallest iteration that keeps the hubs' distance bounds in LDS (edges)"},
    {"MGX_SSSP_BUILD_LIST", "1: list-based queue build"},
    // ---- neighbour-reduce
    {"MGX_NR_SLICED", "0: the unit blocks instead of the long rows by slice of their destinations"},
    {"MGX_NR_SLICES", "number of hot slices (default: by graph size; at most what the id range holds)"},
    {"MGX_NR_FOLD_DEGS", "\"a/b/c\": the fold's tiers (a workgroup / a wave / eight lanes per row) from these degrees"},
    {"MGX_NR_SUBSET", "0: subset frontiers take the general kernel"},
    {"MGX_NR_PARTS", "1 / 2 / 3: timing runs of the short rows only / the long rows only / both"},
    // ---- partitioned BFS (rank engines)
    {"MGX_DIST_SPEC", "0: a host look per level instead of the level plan"},
    {"MGX_DIST_SPARSE_PUSH", "0: every level sweeps the marks"},
    {"MGX_DIST_DECLARE_MUL", "a sweep fills its id list unless the push stored > N x capacity marks (0: always)"},
    {"MGX_DIST_FUSED_MERGE", "0: OR-merge and queue build as two launches"},
    {"MGX_DIST_BUILD_LIST", "1: list-based queue build on the ranks"},
    {"MGX_DIST_PUSH_SPLIT", "1: the parts of a rank's push as separate launches (statistics)"},
    {"MGX_DIST_DEFER", "0: no deferred hot marks on the ranks"},
    {"MGX_DIST_DENSE_DIV", "a level reads the unit blocks when it holds >= 1 / N of the rank's units"},
    {"MGX_DIST_VSHORT", "0 / N: short rows vertex by vertex never / threshold divisor"},
    {"MGX_DIST_COLD", "0: no cold-edge lists per rank"},
    {"MGX_DIST_COLD_WGS", "cold workgroups per rank (measurements)"},
    {"MGX_DIST_COLD_REDUCE", "0: the sweep ORs the cold bitmaps itself"},
    {"MGX_DIST_HOT_UNITS", "0: unit blocks with all entries on the ranks"},
    {"MGX_LOOPBACK_TIMEOUT_S", "deadline of every wait of the loopback communicator (seconds, default 120)"},
  };
  *count = (int)(sizeof(table) / sizeof(table[0]));
  return table;
}

// value of a switch; NULL when unset or empty.  A name that is not in the table is a bug in the caller.
inline const char* env(const char* name) {
  int n = 0;
  const env_switch_t* t = env_switches(&n);
  bool known = false;
  for (int i = 0; i < n && !known; ++i) known = std::strcmp(t[i].name, name) == 0;
  if (!known) {
    std::fprintf(stderr, "mgx: environment switch %s is not in the table of include/mgx/env.hpp\n", name);
    std::abort();
  }
  const char* e = std::getenv(name);
  return (e && *e) ? e : nullptr;
}

}  // namespace mgx
