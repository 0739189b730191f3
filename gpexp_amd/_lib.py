"""ctypes binding of libgpx_hip.so (C ABI in include/gpx.h).

The library is the product: there is no CPU fallback.  `load()` raises ImportError when the
shared object has not been built (run `python -c "import __graft_entry__ as g; g.build()"` or
`make -C gpexp_amd/csrc`), and `Context()` raises RuntimeError when no MI355X is visible.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgpx_hip.so")

c_i64 = C.c_int64
c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(c_i64)
c_vp = C.c_void_p

# name -> (restype, argtypes); mirrors include/gpx.h one to one
_SIGS = {
    "gpx_abi_version": (C.c_int, []),
    "gpx_last_error": (C.c_char_p, []),
    "gpx_create": (C.c_int, [C.c_int, C.POINTER(c_vp)]),
    "gpx_destroy": (C.c_int, [c_vp]),
    "gpx_sync": (C.c_int, [c_vp]),
    "gpx_trim": (C.c_int, [c_vp]),
    "gpx_device_info": (C.c_int, [c_vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), c_ip, C.POINTER(C.c_int)]),
    "gpx_mat_from_host": (C.c_int, [c_vp, c_dp, c_i64, c_i64, C.c_int, C.POINTER(c_vp)]),
    "gpx_mat_alloc": (C.c_int, [c_vp, c_i64, c_i64, C.c_int, C.POINTER(c_vp)]),
    "gpx_mat_free": (C.c_int, [c_vp, c_vp]),
    "gpx_mat_clone": (C.c_int, [c_vp, c_vp, C.POINTER(c_vp)]),
    "gpx_mat_shape": (C.c_int, [c_vp, c_ip, c_ip, c_ip]),
    "gpx_mat_to_host": (C.c_int, [c_vp, c_vp, c_dp, C.c_int]),
    "gpx_kfill": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, c_dp, c_i64, C.POINTER(c_vp)]),
    "gpx_kfill_into": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, c_dp, c_i64, c_vp]),
    "gpx_kdiag": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_dp]),
    "gpx_kernel_eval": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_dp, c_i64, c_dp, c_i64, c_dp]),
    "gpx_potrf": (C.c_int, [c_vp, c_vp]),
    "gpx_potrf_policy": (C.c_int, [c_vp, C.c_double, C.c_int]),
    "gpx_potrf_dropped": (C.c_int, [c_vp, C.POINTER(C.c_int)]),
    "gpx_refit_rows": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_dp, c_i64, c_vp, c_i64, C.POINTER(c_vp)]),
    "gpx_potrs": (C.c_int, [c_vp, c_vp, c_dp, c_dp]),
    "gpx_col_sumsq": (C.c_int, [c_vp, c_vp, c_i64, c_dp]),
    "gpx_dist_ivar_step": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_vp]),
    "gpx_dist_ivar_group": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp]),
    "gpx_dist_ivar_group_at": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64]),
    "gpx_dist_fwd_group_at": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64]),
    "gpx_matvec": (C.c_int, [c_vp, c_vp, c_dp, c_dp]),
    "gpx_fitc_fit": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, C.c_double, C.POINTER(c_vp)]),
    "gpx_fitc_free": (C.c_int, [c_vp, c_vp]),
    "gpx_fitc_shape": (C.c_int, [c_vp, C.POINTER(c_i64), C.POINTER(c_i64)]),
    "gpx_fitc_solve": (C.c_int, [c_vp, c_vp, c_dp, c_dp, c_dp]),
    "gpx_fitc_logdet": (C.c_int, [c_vp, c_vp, c_dp]),
    "gpx_fitc_posterior": (C.c_int, [c_vp, c_vp, c_vp, c_dp, c_vp, c_dp, c_dp]),
    "gpx_fitc_dense": (C.c_int, [c_vp, c_vp, c_dp, c_dp]),
    "gpx_potrs_dev": (C.c_int, [c_vp, c_vp, c_vp, c_vp]),
    "gpx_logdet": (C.c_int, [c_vp, c_vp, c_dp]),
    "gpx_potri": (C.c_int, [c_vp, c_vp, C.POINTER(c_vp)]),
    "gpx_posterior": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, c_dp, c_vp, c_dp, c_dp]),
    "gpx_posterior_cov": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, c_vp, c_dp]),
    "gpx_fit_ivar": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, c_vp, c_dp]),
    "gpx_ivar": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, c_vp, c_dp]),
    "gpx_ivar_keep": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, c_vp, c_dp, C.POINTER(c_vp)]),
    "gpx_ivar_update": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_dp]),
    "gpx_greedy_var": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_dp, c_ip, c_i64, c_i64, c_ip]),
    "gpx_greedy_ivar_step": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, c_vp, c_vp,
                                       C.c_double, c_dp, c_ip]),
    "gpx_greedy_ivar": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, c_vp, c_vp, C.c_double, c_i64, c_ip, c_dp,
                                  c_dp]),
    "gpx_givar_begin": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, c_vp, c_vp, C.c_double, c_i64,
                                  C.POINTER(c_vp)]),
    "gpx_givar_pivot_elems": (c_i64, [c_vp]),
    "gpx_givar_score": (C.c_int, [c_vp, c_vp, c_dp, c_ip, c_dp]),
    "gpx_givar_pack": (C.c_int, [c_vp, c_vp, c_i64, c_vp]),
    "gpx_givar_apply": (C.c_int, [c_vp, c_vp, c_vp]),
    "gpx_givar_end": (C.c_int, [c_vp, c_vp]),
    "gpx_mi_greedy": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, C.c_double, c_i64, c_i64, c_ip, c_dp]),
    "gpx_lml_grad": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, c_dp, c_dp]),
    "gpx_lml_grad_slab": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, c_dp, c_i64, c_i64, c_dp]),
    "gpx_lml_grad_rows": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, c_dp, c_i64, c_i64, C.c_int, c_dp]),
    "gpx_lml_grad_linv": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, c_dp, c_dp]),
    "gpx_mi_begin": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, C.c_double, c_i64, c_i64, c_i64, c_i64, C.POINTER(c_vp)]),
    "gpx_mi_row": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_vp]),
    "gpx_mi_score": (C.c_int, [c_vp, c_vp, c_i64, c_vp, c_dp, c_ip]),
    "gpx_mi_select": (C.c_int, [c_vp, c_vp, c_i64, c_i64]),
    "gpx_mi_end": (C.c_int, [c_vp, c_vp]),
    "gpx_mat_read": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_dp]),
    "gpx_mat_write": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_dp]),
    "gpx_comm_unique_id": (C.c_int, [c_vp]),
    "gpx_comm_init": (C.c_int, [c_vp, C.c_int, C.c_int, c_vp]),
    "gpx_comm_destroy": (C.c_int, [c_vp]),
    "gpx_comm_bcast": (C.c_int, [c_vp, c_vp, c_i64, C.c_int]),
    "gpx_comm_allgather_host": (C.c_int, [c_vp, c_dp, c_i64, c_dp]),
    "gpx_dist_kfill": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_dp, c_i64, c_vp, c_i64, C.c_int,
                                 C.c_int]),
    "gpx_dist_panel_elems": (c_i64, [c_i64, c_i64]),
    "gpx_dist_panel_factor": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_vp]),
    "gpx_dist_begin": (C.c_int, [c_vp]),
    "gpx_dist_info": (C.c_int, [c_vp, C.POINTER(C.c_int)]),
    "gpx_dist_panel_store": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_vp]),
    "gpx_dist_panel_update": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, C.c_int, C.c_int]),
    "gpx_vec_op": (C.c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, C.c_int]),
    "gpx_comm_grid": (C.c_int, [c_vp, C.c_int, C.c_int]),
    "gpx_comm_bcast_grp": (C.c_int, [c_vp, c_vp, c_i64, c_i64, C.c_int, C.c_int]),
    "gpx_comm_bcast_grp2": (C.c_int, [c_vp, c_vp, c_i64, c_vp, c_i64, c_i64, C.c_int, C.c_int]),
    "gpx_comm_reduce_grp": (C.c_int, [c_vp, c_vp, c_i64, c_i64, C.c_int, C.c_int]),
    "gpx_comm_allreduce": (C.c_int, [c_vp, c_vp, c_i64, c_i64]),
    "gpx_comm_allreduce_host": (C.c_int, [c_vp, c_dp, c_i64]),
    "gpx_comm_panel_bcast": (C.c_int, [c_vp, c_vp, c_ip, c_ip, C.POINTER(C.c_int), C.c_int]),
    "gpx_dist2_diag_elems": (c_i64, [c_i64]),
    "gpx_dist2_row_stride": (c_i64, [c_i64]),
    "gpx_dist2_kfill": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_dp, c_i64, c_vp, c_i64, C.c_int, C.c_int,
                                  C.c_int, C.c_int]),
    "gpx_dist2_diag_factor": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64]),
    "gpx_dist2_diag_stage": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64]),
    "gpx_dist2_diag_update": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64]),
    "gpx_dist2_diag_factor_staged": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64]),
    "gpx_dist2_diag_store": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64]),
    "gpx_dist2_reserve": (C.c_int, [c_vp, c_i64, c_i64, c_i64]),
    "gpx_dist2_panel_inv": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64]),
    "gpx_dist2_panel_trsm": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64]),
    "gpx_dist2_panel_trsm_keep": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64]),
    "gpx_dist2_panel_trsm_inv": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64, C.c_int]),
    "gpx_dist2_panel_copyback": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64]),
    "gpx_points_set_box": (C.c_int, [c_vp, c_vp, c_dp, c_dp, C.c_int]),
    "gpx_dist2_panel_pack": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64]),
    "gpx_dist2_diag_pack": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64]),
    "gpx_dist2_update": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64]),
    "gpx_dist2_update_multi": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_i64, C.c_int, C.c_int, C.c_int, C.c_int, c_i64,
                                         C.c_int, C.POINTER(c_vp), c_ip, C.c_int]),
    "gpx_dist2_pack_rows": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_i64]),
    "gpx_dist2_pack_diag": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64]),
    "gpx_program_run": (C.c_int, [c_vp, c_ip, c_i64, c_ip, c_i64, c_dp]),
    "gpx_program_capture": (C.c_int, [c_vp, c_ip, c_i64, c_ip, c_i64, C.POINTER(c_vp)]),
    "gpx_graph_launch": (C.c_int, [c_vp, c_vp, c_dp, c_ip]),
    "gpx_graph_free": (C.c_int, [c_vp, c_vp]),
    "gpx_dist2_unpack_rows": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64]),
    "gpx_dist2_unpack_diag": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64]),
    "gpx_dist2_unpack_diag_at": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64]),
    "gpx_dist2_trsv_diag": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_vp, c_i64, C.c_int]),
    "gpx_dist2_gemv": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, C.c_int]),
    "gpx_dist2_logdet_acc": (C.c_int, [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp]),
    "gpx_stream_select": (C.c_int, [c_vp, C.c_int]),
    "gpx_event_record": (C.c_int, [c_vp, C.c_int]),
    "gpx_event_wait": (C.c_int, [c_vp, C.c_int]),
    "gpx_dist_finish": (C.c_int, [c_vp, c_vp]),
    "gpx_ivar_grad": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, c_vp, c_dp, c_dp]),
    "gpx_ivar_grad_w": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, c_vp, c_dp, c_vp, c_dp]),
    "gpx_ivar_grad_rows": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, c_vp, c_vp, c_i64, c_dp]),
    "gpx_var_grad": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, c_vp, c_dp, c_dp, c_dp, c_dp]),
    "gpx_var_grad_newpt": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, c_vp, c_dp]),
    "gpx_fitc_var_grad": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, c_dp, c_dp, c_dp, c_dp]),
    "gpx_fitc_var_grad_newpt": (C.c_int, [c_vp, c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, c_dp]),
    "gpx_profile_enable": (C.c_int, [c_vp, C.c_int]),
    "gpx_profile_reset": (C.c_int, [c_vp]),
    "gpx_profile_get": (C.c_int, [c_vp, C.c_int, c_ip, c_dp, c_dp, c_dp]),
    "gpx_dbg_gemm": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int]),
    "gpx_dbg_gemm_tri": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int, C.c_int]),
    "gpx_dbg_gemm_ksplit": (C.c_int, [c_vp, c_vp, c_vp, c_vp, C.c_int, C.c_int]),
    "gpx_dbg_kfill_plan": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp, C.c_int, c_vp, c_vp, C.POINTER(C.c_int), c_dp]),
    "gpx_dbg_colreduce_plan": (C.c_int, [c_i64, c_i64, C.POINTER(c_i64), C.POINTER(c_i64)]),
    "gpx_dbg_panel_bcast_plan": (C.c_int, [C.c_int, C.c_int, c_i64, C.c_int, C.POINTER(c_i64), C.POINTER(C.c_int),
                                           C.POINTER(c_i64), c_i64, C.POINTER(c_i64)]),
    "gpx_dbg_guard_violations": (c_i64, [c_vp]),
    "gpx_dbg_guard_selftest": (C.c_int, [c_vp]),
    "gpx_dbg_spin": (C.c_int, [c_vp, C.c_int]),
    "gpx_dbg_spin_us": (C.c_int, [c_vp, c_i64]),
    "gpx_dbg_stamp": (C.c_int, [c_vp, C.c_int]),
    "gpx_dbg_spin_until": (C.c_int, [c_vp, C.c_int, c_i64]),
    "gpx_dbg_event_elapsed": (C.c_int, [c_vp, C.c_int, C.c_int, c_dp]),
    "gpx_dbg_leaf_stamps": (C.c_int, [c_vp, c_vp, C.c_int, C.POINTER(c_i64)]),
}

_lib = None


def load():
    """dlopen the in-tree library and attach signatures; ImportError if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("GPX_LIB_PATH", LIB_PATH)  # developer knob: A/B-test an alternative build of the same ABI
    if not os.path.exists(path):
        raise ImportError(
            "gpexp_amd: %s not found - the HIP library is required (no CPU fallback). "
            "Build it with `make -C gpexp_amd/csrc` or __graft_entry__.build()." % path)
    lib = C.CDLL(path)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.gpx_abi_version() != 2:
        raise ImportError("gpexp_amd: libgpx_hip.so ABI version mismatch")
    _lib = lib
    return lib


def hdot(a, b):
    """sum_i a_i b_i on the host WITHOUT BLAS (NumPy's pairwise sum of the element-wise product).  np.dot on a vector of a few
    thousand entries wakes OpenBLAS' whole thread pool (64 threads on the MI355X boxes), whose workers then spin for a while: in
    a CPU-quota'd container that burns the cgroup's CFS quota and stalls the NEXT fit's ~2000 kernel launches for most of a
    100 ms period -- measured in round 6: a likelihood evaluation at N = 16384 took 98 ms instead of 31 (scripts/bench_frows.py
    f3).  Deterministic, like everything else on the path."""
    import numpy as _np
    return float(_np.sum(_np.multiply(a, b)))


def exported_symbols():
    return sorted(_SIGS)


class GpxError(RuntimeError):
    pass


class NotPositiveDefinite(GpxError):
    def __init__(self, pivot):
        super().__init__("covariance matrix is not positive definite: pivot %d <= 0 "
                         "(the reference's pinv would silently truncate here, gp.py:181)" % pivot)
        self.pivot = pivot


def check(rc):
    if rc == 0:
        return
    if rc > 0:
        raise NotPositiveDefinite(rc)
    raise GpxError("libgpx_hip: " + load().gpx_last_error().decode("utf-8", "replace"))


def as_f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def dptr(a):
    return a.ctypes.data_as(c_dp) if a is not None else None
