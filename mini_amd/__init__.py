"""mini_amd -- MI355X-native frontier traversal engine (advance / filter / neighbourhood-reduce and
the BFS / SSSP / PR loops that drive them), behind mini-gunrock's operator API.

Layout: csrc/ (HIP sources of libmgx.so), _lib.py (ctypes binding of include/mgx.h),
api.py (host mirror of the reference's data model), rmat.py (synthetic inputs, device-side).
Importing the package loads libmgx.so and raises ImportError if it has not been built.
"""
from ._lib import (LIB_PATH, MGX_BFS_DIRECTION_OPT, MGX_BFS_PUSH, MGX_E_FRONTIER_OVERFLOW, MGX_E_INVALID,
                   MGX_E_NEGATIVE_WEIGHT, MgxError, lib)
from .api import (BfsProblem, Context, Frontier, Graph, KcoreProblem, PrProblem, SsspProblem, compact_i32, lbs_expand_debug,
                  load_csr_cache, load_mtx, rmat_edges, save_csr_cache, scan_exclusive_i32, scan_frontier_degrees, segreduce)

__all__ = [n for n in dir() if not n.startswith("_")]
