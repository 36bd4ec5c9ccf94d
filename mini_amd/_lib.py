"""ctypes binding of libmgx.so (the C-ABI declared in include/mgx.h).

The library is the product: importing this module FAILS LOUDLY when it is missing -- there is
no Python/CPU fallback for any operator.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MGX_LIB", os.path.join(_HERE, "libmgx.so"))

MGX_OK = 0
MGX_E_INVALID = -1
MGX_E_HIP = -2
MGX_E_FRONTIER_OVERFLOW = -4
MGX_E_NEGATIVE_WEIGHT = -5
MGX_E_NO_DEVICE = -6
MGX_BFS_PUSH = 0
MGX_BFS_DIRECTION_OPT = 1


class MgxError(RuntimeError):
    def __init__(self, status, detail):
        super().__init__("mgx status %d: %s" % (status, detail))
        self.status = status


def _load():
    # When PyTorch is in the process, let it load ITS bundled HIP runtime first: libmgx.so needs
    # libamdhip64.so.7 and then binds to the copy already loaded instead of bringing /opt/rocm's
    # second runtime into the process (two runtimes => the second one sees no device).
    if os.environ.get("MGX_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
            if torch.cuda.is_available():
                torch.cuda.init()
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libmgx.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no fallback path." % LIB_PATH)
    return C.CDLL(LIB_PATH)


lib = _load()

_vp, _i, _i64, _f, _u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint64
_pi, _pi64, _pf, _pvp = C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_float), C.POINTER(C.c_void_p)

# name -> argtypes (restype is int unless listed in _RESTYPES)
SIGNATURES = {
    "mgx_version": [],
    "mgx_build_is_lab": [],
    "mgx_env_switches": [_i, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)],
    "mgx_strerror": [_i],
    "mgx_last_error": [],
    "mgx_ctx_create": [_i, _vp, _pvp],
    "mgx_ctx_set_stream": [_vp, _vp],
    "mgx_ctx_synchronize": [_vp],
    "mgx_ctx_destroy": [_vp],
    "mgx_ctx_num_cus": [_vp, _pi],
    "mgx_graph_upload": [_vp, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _pvp],
    "mgx_graph_wrap_device": [_vp, _i, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _pvp],
    "mgx_graph_attach_layout": [_vp, _vp, _vp, _vp, _vp],
    "mgx_graph_attach_layout_weights": [_vp, _vp],
    "mgx_graph_build_layout": [_vp, _i],
    "mgx_graph_build_csc": [_vp],
    "mgx_graph_csc_read": [_vp, _vp, _vp, _vp],
    "mgx_graph_layout_read": [_vp, _vp, _vp, _vp, _vp, _vp],
    "mgx_graph_layout_info": [_vp, _pi64],
    "mgx_graph_nr_slices_info": [_vp, _pi64],
    "mgx_graph_free": [_vp],
    "mgx_graph_dims": [_vp, _pi, _pi64],
    "mgx_load_mtx": [C.c_char_p, _i, _i, _pi, _pi64, C.POINTER(_pi), C.POINTER(_pi), C.POINTER(_pf)],
    "mgx_graph_save_csr": [C.c_char_p, _i, _i64, _i, _vp, _vp, _vp, _vp, _vp, _vp],
    "mgx_graph_load_csr": [C.c_char_p, _pi, _pi64, _pi, C.POINTER(_pi), C.POINTER(_pi), C.POINTER(_pf), C.POINTER(_pi),
                           C.POINTER(_pi), C.POINTER(_pf)],
    "mgx_load_mtx_csc": [C.c_char_p, _i, _i, _i, _pi, _pi64, C.POINTER(_pi), C.POINTER(_pi), C.POINTER(_pf),
                         C.POINTER(_pi), C.POINTER(_pi), C.POINTER(_pf)],
    "mgx_host_free": [_vp],
    "mgx_frontier_create": [_vp, _i64, _pvp],
    "mgx_frontier_free": [_vp],
    "mgx_frontier_load": [_vp, _vp, _i64],
    "mgx_frontier_fill_iota": [_vp, _i64],
    "mgx_frontier_fill": [_vp, _i, _i64],
    "mgx_frontier_read": [_vp, _vp, _i64, _pi64],
    "mgx_frontier_resize": [_vp, _i64],
    "mgx_frontier_size": [_vp, _pi64],
    "mgx_frontier_capacity": [_vp, _pi64],
    "mgx_frontier_device_ptr": [_vp, _pvp],
    "mgx_scan_exclusive_i32": [_vp, _vp, _i64, _vp, _pi64],
    "mgx_scan_frontier_degrees": [_vp, _vp, _i, _pi64],
    "mgx_lbs_expand_debug": [_vp, _vp, _i64, _vp, _vp],
    "mgx_compact_i32": [_vp, _vp, _i64, _i, _vp, _pi64],
    "mgx_segreduce_f32_plus": [_vp, _vp, _i, _vp, _f, _vp, _pi64],
    "mgx_segreduce_i32_min": [_vp, _vp, _i, _vp, _i, _vp, _pi64],
    "mgx_segreduce_i32_max": [_vp, _vp, _i, _vp, _i, _vp, _pi64],
    "mgx_bfs_create": [_vp, _i, _pvp],
    "mgx_bfs_reset": [_vp, _i],
    "mgx_bfs_free": [_vp],
    "mgx_bfs_labels": [_vp, _vp],
    "mgx_bfs_preds": [_vp, _vp],
    "mgx_bfs_labels_device": [_vp, _pvp],
    "mgx_bfs_advance": [_vp, _vp, _vp, _i, _pi64],
    "mgx_bfs_filter": [_vp, _vp, _vp, _i, _pi64],
    "mgx_bfs_advance_filter_fused": [_vp, _vp, _vp, _i, _pi64],
    "mgx_bfs_gen_unvisited": [_vp, _vp, _vp, _i, _pi64],
    "mgx_bfs_sparse_to_dense": [_vp, _vp, _vp, _i],
    "mgx_bfs_advance_backward": [_vp, _vp, _vp, _vp, _i, _pi64],
    "mgx_bfs_enact_pushpull": [_vp, _f, _pi64],
    "mgx_bfs_advance_idempotent": [_vp, _vp, _vp, _i, _pi64],
    "mgx_bfs_uniquify": [_vp, _vp, _vp, _i, _pi64],
    "mgx_bfs_enact_idempotent": [_vp, _pi64],
    "mgx_bfs_run": [_vp, _i, _i, _f, _pi64],
    "mgx_bfs_run_stats": [_vp, _i, _i, _f, _pi64, _i],
    "mgx_bfs_run_many": [_vp, _pi, _i, _i, _f, _pi64, _i, _pi],
    "mgx_bfs_level_trace": [_vp, _i, _pi64, _pi64, _pi],
    "mgx_bfs_set_kernel_timing": [_vp, _i],
    "mgx_bfs_kernel_times": [_vp, _pi64],
    "mgx_bfs_level_kernel_times": [_vp, _i, _pf, _pf],
    "mgx_bfs_level_times": [_vp, _i, _pf, _pi],
    "mgx_bfs_level_claims": [_vp, _i, _pi64],
    "mgx_bfs_batch_times": [_vp, _i, _pf, _pi],
    "mgx_dbfs_create": [_vp, _i, _i, _i, _i64, _vp, _vp, _vp, _i64, _pvp],
    "mgx_dbfs_free": [_vp],
    "mgx_dbfs_range": [_vp, _pi, _pi],
    "mgx_dbfs_reset": [_vp, _i],
    "mgx_dbfs_expand": [_vp, _pi64, _pi64],
    "mgx_dbfs_bins": [_vp, _pvp, _pi64],
    "mgx_dbfs_receive": [_vp, _vp, _i64, _i],
    "mgx_dbfs_swap": [_vp, _pi64],
    "mgx_dbfs_labels": [_vp, _vp],
    "mgx_comm_unique_id": [_vp],
    "mgx_comm_create": [_vp, _i, _i, _vp, _pvp],
    "mgx_comm_free": [_vp],
    "mgx_comm_library": [],
    "mgx_comm_available": [],
    "mgx_comm_loopback_id": [_vp],
    "mgx_sssp_build_preds": [_vp, _pi64],
    "mgx_dbfs2_spec_stats": [_vp, _pi64],
    "mgx_dbfs2_forget_plan": [_vp],
    "mgx_dbfs2_run_group": [_vp, _i, _i, _i64, _pi64],
    "mgx_comm_info": [_vp, _vp, _vp],
    "mgx_comm_selftest": [_vp, _vp, _vp, _vp, _i64],
    "mgx_dbfs2_run": [_vp, _vp, _i, _i, _i64, _pi64],
    "mgx_dbfs2_build_units": [_vp, _pi64],
    "mgx_dbfs2_dense_levels": [_vp, _pi64],
    "mgx_dbfs2_cold_levels": [_vp, _pi64, _pi64],
    "mgx_dbfs2_path_levels": [_vp, _pi64],
    "mgx_dsssp_create": [_vp, _i, _i, _i, _i64, _vp, _vp, _vp, _pvp],
    "mgx_dsssp_free": [_vp],
    "mgx_dsssp_reset": [_vp, _i],
    "mgx_dsssp_expand": [_vp, _pi64, _pi64],
    "mgx_dsssp_bins": [_vp, _pvp, _pi64],
    "mgx_dsssp_receive": [_vp, _vp, _i64],
    "mgx_dsssp_swap": [_vp, _pi64],
    "mgx_dsssp_distances": [_vp, _vp],
    "mgx_dsssp_run": [_vp, _vp, _i, _pi64],
    "mgx_dbfs2_create": [_vp, _i, _i, _i, _vp, _vp, _vp, _pvp],
    "mgx_dbfs2_free": [_vp],
    "mgx_dbfs2_reset": [_vp, _i],
    "mgx_dbfs2_words": [_i, _pi64],
    "mgx_dbfs2_status": [_vp, _i, _pi64],
    "mgx_dbfs2_shard_plan": [_vp, _i, _i, _u64, _i, _i, _pvp, _pi, _pi64],
    "mgx_dbfs2_shard_fill": [_vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "mgx_dbfs2_shard_free": [_vp],
    "mgx_dbfs2_list_words": [_i, _i, _pi64],
    "mgx_dbfs2_set_list": [_vp, _vp, _i64],
    "mgx_dbfs2_apply_lists": [_vp, _i, _vp, _i, _i64, _pi64],
    "mgx_dbfs2_push": [_vp, _i],
    "mgx_dbfs2_merge": [_vp, _i, _vp],
    "mgx_dbfs2_merge_maps": [_vp, _i, _vp, _i, _i64],
    "mgx_dbfs2_or_maps": [_vp, _vp, _i, _i64, _i64, _vp],
    "mgx_dbfs2_labels": [_vp, _vp],
    "mgx_dbfs2_visited": [_vp, _vp],
    "mgx_sssp_create": [_vp, _i, _pvp],
    "mgx_sssp_reset": [_vp, _i],
    "mgx_sssp_free": [_vp],
    "mgx_sssp_distances": [_vp, _vp],
    "mgx_sssp_preds": [_vp, _vp],
    "mgx_sssp_distances_device": [_vp, _pvp],
    "mgx_sssp_advance": [_vp, _vp, _vp, _i, _pi64],
    "mgx_sssp_filter": [_vp, _vp, _vp, _i, _pi64],
    "mgx_sssp_enact": [_vp, _f, _pi64],
    "mgx_sssp_run": [_vp, _i, _pi64],
    "mgx_sssp_run_delta": [_vp, _i, _f, _pi64],
    "mgx_sssp_set_kernel_timing": [_vp, _i],
    "mgx_sssp_iteration_trace": [_vp, _i, _pi64, _pi64, _pf, _pi],
    "mgx_sssp_kernel_times": [_vp, _pi64],
    "mgx_pr_create": [_vp, _i, _pvp],
    "mgx_pr_free": [_vp],
    "mgx_pr_enact": [_vp, _pi64, _pi],
    "mgx_pr_ranks": [_vp, _vp],
    "mgx_kcore_create": [_vp, _pvp],
    "mgx_kcore_reset": [_vp],
    "mgx_kcore_free": [_vp],
    "mgx_kcore_enact": [_vp, _pi, _pi64],
    "mgx_kcore_num_cores": [_vp, _vp],
    "mgx_kcore_degrees": [_vp, _vp],
    "mgx_rmat_edges": [_vp, _i, _i64, _i64, _u64, _i, _vp, _vp, _vp],
}
_RESTYPES = {"mgx_comm_library": C.c_char_p, "mgx_strerror": C.c_char_p, "mgx_last_error": C.c_char_p, "mgx_host_free": None}

for _name, _args in SIGNATURES.items():
    _fn = getattr(lib, _name)          # AttributeError here == header/library drift
    _fn.argtypes = _args
    _fn.restype = _RESTYPES.get(_name, C.c_int)


def check(status):
    if status != MGX_OK:
        detail = lib.mgx_last_error()
        raise MgxError(status, (detail or b"").decode() or lib.mgx_strerror(status).decode())
    return status
