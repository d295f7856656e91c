"""ctypes binding of the C ABI declared in include/qpalm_gfx950.h.

`load()` opens the shipped HIP library (qpalm_amd/lib/libqpalm_gfx950.so) and raises if it is
missing -- there is no CPU fallback in the product path.  Tests that exercise the kernel source on
the host pass the path of the test-only emulation build explicitly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libqpalm_gfx950.so")

c_int = C.c_int64
c_float = C.c_double
pf = C.POINTER(c_float)
pi = C.POINTER(c_int)


class Settings(C.Structure):
    """QPALMSettings (include/types.h:119-150; interfaces/python/qpalm.py:49-80)."""
    _fields_ = [
        ("max_iter", c_int), ("inner_max_iter", c_int), ("eps_abs", c_float), ("eps_rel", c_float),
        ("eps_abs_in", c_float), ("eps_rel_in", c_float), ("rho", c_float), ("eps_prim_inf", c_float),
        ("eps_dual_inf", c_float), ("theta", c_float), ("delta", c_float), ("sigma_max", c_float),
        ("sigma_init", c_float), ("proximal", c_int), ("gamma_init", c_float), ("gamma_upd", c_float),
        ("gamma_max", c_float), ("scaling", c_int), ("nonconvex", c_int), ("verbose", c_int),
        ("print_iter", c_int), ("warm_start", c_int), ("reset_newton_iter", c_int),
        ("enable_dual_termination", c_int), ("dual_objective_limit", c_float), ("time_limit", c_float),
        ("ordering", c_int), ("factorization_method", c_int), ("max_rank_update", c_int),
        ("max_rank_update_fraction", c_float),
    ]


class Info(C.Structure):
    """QPALMInfo (include/types.h:76-95; interfaces/python/qpalm.py:98-110)."""
    _fields_ = [
        ("iter", c_int), ("iter_out", c_int), ("status", C.c_char * 32), ("status_val", c_int),
        ("pri_res_norm", c_float), ("dua_res_norm", c_float), ("dua2_res_norm", c_float),
        ("objective", c_float), ("dual_objective", c_float),
        ("setup_time", c_float), ("solve_time", c_float), ("run_time", c_float),
    ]


class Stats(C.Structure):
    _fields_ = [(k, c_int) for k in ("n_refactor", "n_factor_Q", "n_sweeps", "n_rank1", "n_solve", "n_sigma_updates",
                                     "n_boost_gamma", "nb_active", "nb_enter", "nb_leave", "last_kind", "last_fact")] + \
               [(k, c_float) for k in ("gamma", "tau", "eta", "beta", "eps_pri", "eps_dua", "eps_dua_in", "sc_c",
                                       "ms_total", "ms_factor", "ms_update", "ms_solve", "ms_linesearch")] + \
               [("ms_dbg", c_float * 16), ("sweep_entries", c_int), ("factor_reread_entries", c_int),
                ("lobpcg_lambda", c_float), ("placement", c_int), ("lobpcg_iter", c_int), ("nonconvex", c_int), ("n_fused_solve", c_int),
                ("n_seq_columns", c_int), ("n_sweep_columns", c_int), ("n_guard_refactor", c_int)]


class QpgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("qpalm_gfx950 error %d: %s" % (code, msg))
        self.code = code


_LIBS = {}


def load(path=None):
    path = path or LIB_PATH
    if path in _LIBS:
        return _LIBS[path]
    if not os.path.exists(path):
        raise ImportError("%s is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; "
                          "g.build()'); there is no CPU fallback" % path)
    L = C.CDLL(path)
    L.qpg_last_error.restype = C.c_char_p
    L.qpg_backend_name.restype = C.c_char_p
    L.qpg_set_default_settings.argtypes = [C.POINTER(Settings)]
    L.qpg_validate_settings.argtypes = [C.POINTER(Settings)]
    L.qpg_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.qpg_ctx_destroy.argtypes = [C.c_void_p]
    L.qpg_ctx_destroy.restype = None
    L.qpg_ctx_set_option.argtypes = [C.c_void_p, C.c_char_p, c_int]
    L.qpg_batch_create.argtypes = [C.c_void_p, c_int, c_int, c_int, c_int, c_int, C.POINTER(Settings), C.POINTER(C.c_void_p)]
    L.qpg_batch_set_problem.argtypes = [C.c_void_p, c_int, pi, pi, pf, pi, pi, pf, pf, c_float, pf, pf]
    L.qpg_batch_set_problem_sized.argtypes = [C.c_void_p, c_int, c_int, c_int, pi, pi, pf, pi, pi, pf, pf, c_float, pf, pf]
    pp = C.POINTER(C.c_void_p)
    L.qpg_batch_set_problems.argtypes = [C.c_void_p, c_int, c_int, pi, pi, pp, pp, pp, pp, pp, pp, pp, pf, pp, pp]
    for f in ("qpg_batch_setup", "qpg_batch_solve", "qpg_batch_sync", "qpg_batch_begin_solve", "qpg_batch_warm_start_last"):
        getattr(L, f).argtypes = [C.c_void_p]
    L.qpg_batch_warm_start.argtypes = [C.c_void_p, pf, pf]
    L.qpg_batch_iterate.argtypes = [C.c_void_p, c_int]
    L.qpg_batch_num_unfinished.argtypes = [C.c_void_p, pi]
    L.qpg_batch_launch_shape.argtypes = [C.c_void_p, pi, pi, pi]
    if hasattr(L, "qpg_batch_sparse_info"):
        L.qpg_batch_sparse_info.argtypes = [C.c_void_p, c_int, pi, pi]
    if hasattr(L, "qpg_batch_sparse_perm"):   # (absent from older builds of the library that tools/evidence/gpu_ab.sh compares with)
        L.qpg_batch_sparse_perm.argtypes = [C.c_void_p, c_int, pi, pi]
    L.qpg_batch_last_solve_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    L.qpg_batch_update_settings.argtypes = [C.c_void_p, C.POINTER(Settings)]
    L.qpg_batch_update_bounds.argtypes = [C.c_void_p, pf, pf]
    L.qpg_batch_update_q.argtypes = [C.c_void_p, pf]
    L.qpg_batch_get_info.argtypes = [C.c_void_p, c_int, C.POINTER(Info)]
    L.qpg_batch_get_stats.argtypes = [C.c_void_p, c_int, C.POINTER(Stats)]
    L.qpg_batch_get_info_all.argtypes = [C.c_void_p, C.POINTER(Info)]
    L.qpg_batch_get_stats_all.argtypes = [C.c_void_p, C.POINTER(Stats)]
    L.qpg_batch_get_solution.argtypes = [C.c_void_p, pf, pf]
    L.qpg_batch_get_vector.argtypes = [C.c_void_p, C.c_char_p, c_int, pf, c_int]
    L.qpg_batch_set_vector.argtypes = [C.c_void_p, C.c_char_p, c_int, pf, c_int]
    L.qpg_batch_get_ivector.argtypes = [C.c_void_p, C.c_char_p, c_int, pi, c_int]
    L.qpg_batch_set_ivector.argtypes = [C.c_void_p, C.c_char_p, c_int, pi, c_int]
    L.qpg_batch_set_scalar.argtypes = [C.c_void_p, C.c_char_p, c_int, c_float]
    L.qpg_batch_get_factor.argtypes = [C.c_void_p, c_int, pf, pf, c_int]
    L.qpg_batch_destroy.argtypes = [C.c_void_p]
    L.qpg_batch_destroy.restype = None
    L.qpg_batch_device_ptr.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.qpg_mat_vec.argtypes = [C.c_void_p, c_int, C.c_int, pf, pf]
    L.qpg_mat_tpose_vec.argtypes = [C.c_void_p, c_int, C.c_int, pf, pf]
    for f in ("qpg_ldlchol", "qpg_ldlcholQAtsigmaA", "qpg_ldlupdate_entering_constraints",
              "qpg_ldldowndate_leaving_constraints", "qpg_ldlupdate_sigma_changed", "qpg_ldlsolveLD_neg_dphi",
              "qpg_compute_residuals", "qpg_set_active_constraints", "qpg_kkt_form", "qpg_kkt_factorize",
              "qpg_kkt_update_entering_constraints", "qpg_kkt_update_leaving_constraints", "qpg_kkt_solve"):
        getattr(L, f).argtypes = [C.c_void_p, c_int]
    L.qpg_exact_linesearch.argtypes = [C.c_void_p, c_int, pf]
    L.qpg_ldlchol_matrix.argtypes = [C.c_void_p, c_int, c_int, pi, pi, pf]
    L.qpg_sparse_matvec.argtypes = [C.c_void_p, c_int, c_int, pi, pi, pf, C.c_int, C.c_int, pf, pf]
    L.qpg_batch_ldlsolve_all.argtypes = [C.c_void_p, c_int, C.POINTER(C.c_float)]
    if hasattr(L, "qpg_batch_sweep_probe"):
        L.qpg_batch_sweep_probe.argtypes = [C.c_void_p, c_int, c_int, C.POINTER(C.c_float)]
    L.qpg_ctx_hbm_copy_gbs.argtypes = [C.c_void_p, C.c_size_t, c_int, C.POINTER(C.c_float)]
    L.qpg_ctx_hbm_read_gbs.argtypes = [C.c_void_p, C.c_size_t, c_int, C.POINTER(C.c_float)]
    L.qpg_host_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
    L.qpg_host_free.argtypes = [C.c_void_p, C.c_void_p]
    _LIBS[path] = L
    return L


# every symbol include/qpalm_gfx950.h declares (checked by tests/test_capi_symbols.py)
SYMBOLS = [
    "qpg_last_error", "qpg_backend_name", "qpg_set_default_settings", "qpg_validate_settings", "qpg_ctx_create",
    "qpg_ctx_destroy", "qpg_ctx_set_option", "qpg_batch_create", "qpg_batch_set_problem", "qpg_batch_setup",
    "qpg_batch_warm_start", "qpg_batch_warm_start_last", "qpg_batch_solve", "qpg_batch_iterate", "qpg_batch_last_solve_ms", "qpg_batch_num_unfinished", "qpg_batch_launch_shape", "qpg_batch_sparse_info", "qpg_batch_sparse_perm",
    "qpg_batch_update_settings", "qpg_batch_update_bounds", "qpg_batch_update_q", "qpg_batch_get_info",
    "qpg_batch_get_stats", "qpg_batch_get_solution", "qpg_batch_get_vector", "qpg_batch_set_vector",
    "qpg_batch_get_ivector", "qpg_batch_set_ivector", "qpg_batch_set_scalar", "qpg_batch_get_factor",
    "qpg_batch_destroy", "qpg_batch_device_ptr", "qpg_batch_sync", "qpg_mat_vec", "qpg_mat_tpose_vec", "qpg_ldlchol",
    "qpg_ldlcholQAtsigmaA", "qpg_ldlupdate_entering_constraints", "qpg_ldldowndate_leaving_constraints",
    "qpg_ldlupdate_sigma_changed", "qpg_ldlsolveLD_neg_dphi", "qpg_compute_residuals", "qpg_set_active_constraints",
    "qpg_exact_linesearch", "qpg_batch_ldlsolve_all", "qpg_batch_sweep_probe", "qpg_kkt_form", "qpg_kkt_factorize",
    "qpg_kkt_update_entering_constraints", "qpg_kkt_update_leaving_constraints", "qpg_kkt_solve", "qpg_ldlchol_matrix", "qpg_sparse_matvec",
    "qpg_batch_begin_solve", "qpg_batch_get_info_all", "qpg_batch_get_stats_all", "qpg_ctx_hbm_copy_gbs", "qpg_ctx_hbm_read_gbs", "qpg_host_alloc", "qpg_host_free", "qpg_batch_set_problem_sized", "qpg_batch_set_problems",
]


def fptr(a):
    return a.ctypes.data_as(pf)


def iptr(a):
    return a.ctypes.data_as(pi)


def f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def i64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64))
