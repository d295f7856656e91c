"""ctypes binding of the CPU oracle (oracle/libqpalm_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module
(see oracle/qpalm_oracle.h).  The API mirrors the reference's ctypes wrapper
(interfaces/python/qpalm.py:192-375) so that tests read like the reference's own.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_int = C.c_int64
c_float = C.c_double


class Settings(C.Structure):
    """QPALMSettings, include/types.h:119-150 (field order witnessed by interfaces/python/qpalm.py:49-80)."""
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
    """QPALMInfo with PROFILING, include/types.h:76-95."""
    _fields_ = [
        ("iter", c_int), ("iter_out", c_int), ("status", C.c_char * 32), ("status_val", c_int),
        ("pri_res_norm", c_float), ("dua_res_norm", c_float), ("dua2_res_norm", c_float),
        ("objective", c_float), ("dual_objective", c_float),
        ("setup_time", c_float), ("solve_time", c_float), ("run_time", c_float),
    ]


class Sparse(C.Structure):
    _fields_ = [("nrow", c_int), ("ncol", c_int), ("nzmax", c_int), ("p", C.POINTER(c_int)),
                ("i", C.POINTER(c_int)), ("x", C.POINTER(c_float)), ("stype", C.c_int)]


class Trace(C.Structure):
    _fields_ = [("cap", c_int), ("len", c_int), ("kind", C.POINTER(c_int)), ("fact", C.POINTER(c_int)),
                ("nb_active", C.POINTER(c_int)), ("nb_enter", C.POINTER(c_int)), ("nb_leave", C.POINTER(c_int)),
                ("tau", C.POINTER(c_float)), ("gamma", C.POINTER(c_float)), ("pri_res_norm", C.POINTER(c_float)),
                ("dua_res_norm", C.POINTER(c_float)), ("dua2_res_norm", C.POINTER(c_float)),
                ("x", C.POINTER(c_float)), ("y", C.POINTER(c_float)), ("d", C.POINTER(c_float)),
                ("active", C.POINTER(c_int))]


def build(force=False):
    so = os.path.join(_HERE, "libqpalm_oracle.so")
    src = os.path.join(_HERE, "qpalm_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def lib(path=None):
    global _LIB
    if path is not None:
        return _declare(C.CDLL(path))
    if _LIB is None:
        _LIB = _declare(C.CDLL(build()))
    return _LIB


def _declare(L):
    pf, pi = C.POINTER(c_float), C.POINTER(c_int)
    L.oq_setup.restype = C.c_void_p
    L.oq_setup.argtypes = [c_int, c_int, pi, pi, pf, pi, pi, pf, pf, c_float, pf, pf, C.POINTER(Settings)]
    L.oq_set_default_settings.argtypes = [C.POINTER(Settings)]
    L.oq_warm_start.argtypes = [C.c_void_p, pf, pf]
    for f in ("oq_solve", "oq_cleanup", "oq_compute_residuals", "oq_set_active_constraints",
              "oq_set_entering_leaving_constraints", "oq_newton_set_direction", "oq_update_primal_iterate",
              "oq_update_sigma", "oq_ldlcholQAtsigmaA", "oq_ldlupdate_entering_constraints",
              "oq_ldldowndate_leaving_constraints", "oq_ldlupdate_sigma_changed", "oq_ldlsolveLD_neg_dphi"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = None
    L.oq_check_termination.argtypes = [C.c_void_p]
    L.oq_check_termination.restype = c_int
    L.oq_exact_linesearch.argtypes = [C.c_void_p]
    L.oq_exact_linesearch.restype = c_float
    L.oq_update_settings.argtypes = [C.c_void_p, C.POINTER(Settings)]
    L.oq_update_bounds.argtypes = [C.c_void_p, pf, pf]
    L.oq_update_q.argtypes = [C.c_void_p, pf]
    L.oq_set_trace.argtypes = [C.c_void_p, C.POINTER(Trace)]
    L.oq_get_info.restype = C.POINTER(Info)
    L.oq_get_info.argtypes = [C.c_void_p]
    L.oq_get_settings.restype = C.POINTER(Settings)
    L.oq_get_settings.argtypes = [C.c_void_p]
    L.oq_get_solution_x.restype = pf
    L.oq_get_solution_x.argtypes = [C.c_void_p]
    L.oq_get_solution_y.restype = pf
    L.oq_get_solution_y.argtypes = [C.c_void_p]
    L.oq_get_vec.restype = pf
    L.oq_get_vec.argtypes = [C.c_void_p, C.c_char_p, pi]
    L.oq_get_ivec.restype = pi
    L.oq_get_ivec.argtypes = [C.c_void_p, C.c_char_p, pi]
    L.oq_get_scalar.restype = c_float
    L.oq_get_scalar.argtypes = [C.c_void_p, C.c_char_p]
    L.oq_set_scalar.argtypes = [C.c_void_p, C.c_char_p, c_float]
    L.oq_get_counter.restype = c_int
    L.oq_get_counter.argtypes = [C.c_void_p, C.c_char_p]
    L.oq_get_matrix.argtypes = [C.c_void_p, C.c_char_p, pi, pi, C.POINTER(pi), C.POINTER(pi), C.POINTER(pf)]
    L.oq_get_factor.restype = pf
    L.oq_get_factor.argtypes = [C.c_void_p, C.POINTER(pf), pi]
    L.oq_set_perm.restype = c_int
    L.oq_set_perm.argtypes = [C.c_void_p, pi, c_int]
    L.oq_sparse_levels.restype = c_int
    L.oq_sparse_levels.argtypes = [C.c_void_p]
    L.oq_get_kkt_factor.restype = pf
    L.oq_get_kkt_factor.argtypes = [C.c_void_p, C.POINTER(pf), pi]
    for f in ("oq_kkt_form_and_factor", "oq_kkt_update_entering_constraints", "oq_kkt_update_leaving_constraints", "oq_kkt_solve"):
        getattr(L, f).argtypes = [C.c_void_p]
        getattr(L, f).restype = None
    L.oq_mat_vec.argtypes = [C.POINTER(Sparse), pf, pf]
    L.oq_mat_tpose_vec.argtypes = [C.POINTER(Sparse), pf, pf]
    L.oq_mat_inf_norm_cols.argtypes = [C.POINTER(Sparse), pf]
    L.oq_mat_inf_norm_rows.argtypes = [C.POINTER(Sparse), pf]
    L.oq_ldlchol.argtypes = [C.POINTER(Sparse), C.c_void_p]
    L.oq_dense_ldl_factor.argtypes = [c_int, pf, c_int, pf]
    L.oq_dense_ldl_solve.argtypes = [c_int, pf, c_int, pf, pf]
    L.oq_dense_ldl_rank1.argtypes = [c_int, pf, c_int, pf, pf, C.c_int]
    L.oq_vec_prod.restype = c_float
    L.oq_vec_prod.argtypes = [pf, pf, C.c_size_t]
    L.oq_vec_norm_inf.restype = c_float
    L.oq_vec_norm_inf.argtypes = [pf, C.c_size_t]
    L.oq_vec_set_scalar.argtypes = [pf, c_float, C.c_size_t]
    L.oq_vec_self_mult_scalar.argtypes = [pf, c_float, C.c_size_t]
    L.oq_vec_add_scaled.argtypes = [pf, pf, pf, c_float, C.c_size_t]
    L.oq_vec_mult_add_scaled.argtypes = [pf, pf, c_float, c_float, C.c_size_t]
    for f in ("oq_vec_ew_recipr", "oq_vec_ew_sqrt"):
        getattr(L, f).argtypes = [pf, pf, C.c_size_t]
    for f in ("oq_vec_ew_max_vec", "oq_vec_ew_min_vec", "oq_vec_ew_prod", "oq_vec_ew_div"):
        getattr(L, f).argtypes = [pf, pf, pf, C.c_size_t]
    L.oq_vec_ew_mid_vec.argtypes = [pf, pf, pf, pf, C.c_size_t]
    return L


def fptr(a):
    return a.ctypes.data_as(C.POINTER(c_float))


def iptr(a):
    return a.ctypes.data_as(C.POINTER(c_int))


def f64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def i64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.int64))


def default_settings(**kw):
    s = Settings()
    lib().oq_set_default_settings(C.byref(s))
    for k, v in kw.items():
        if not hasattr(s, k):
            raise AttributeError(k)
        setattr(s, k, v)
    return s


def make_sparse(nrow, ncol, p, i, x, stype):
    """Returns (Sparse struct, keepalive tuple)."""
    p, i, x = i64(p), i64(i), f64(x)
    S = Sparse(nrow, ncol, len(x), iptr(p), iptr(i), fptr(x), stype)
    return S, (p, i, x)


class OracleQP:
    """One QP workspace of the oracle.  Q is CSC with only the lower triangle read (stype -1)."""

    def __init__(self, n, m, Qp, Qi, Qx, Ap, Ai, Ax, q, bmin, bmax, c=0.0, settings=None, libpath=None):
        self.L = lib(libpath)
        self.n, self.m = int(n), int(m)
        self._keep = [i64(Qp), i64(Qi), f64(Qx), i64(Ap), i64(Ai), f64(Ax), f64(q), f64(bmin), f64(bmax)]
        k = self._keep
        self.settings = settings if settings is not None else default_settings()
        self.w = self.L.oq_setup(self.n, self.m, iptr(k[0]), iptr(k[1]), fptr(k[2]), iptr(k[3]), iptr(k[4]),
                                 fptr(k[5]), fptr(k[6]), float(c), fptr(k[7]), fptr(k[8]), C.byref(self.settings))
        self._trace = None

    @property
    def ok(self):
        return bool(self.w)

    def warm_start(self, x=None, y=None):
        xs = f64(x) if x is not None else None
        ys = f64(y) if y is not None else None
        self.L.oq_warm_start(self.w, fptr(xs) if xs is not None else None, fptr(ys) if ys is not None else None)

    def solve(self):
        self.L.oq_solve(self.w)
        return self.info

    def update_settings(self, s):
        self.L.oq_update_settings(self.w, C.byref(s))

    def update_bounds(self, bmin=None, bmax=None):
        a = f64(bmin) if bmin is not None else None
        b = f64(bmax) if bmax is not None else None
        self.L.oq_update_bounds(self.w, fptr(a) if a is not None else None, fptr(b) if b is not None else None)

    def update_q(self, q):
        q = f64(q)
        self.L.oq_update_q(self.w, fptr(q))

    @property
    def info(self):
        return self.L.oq_get_info(self.w).contents

    @property
    def status_val(self):
        return int(self.info.status_val)

    @property
    def x(self):
        return np.ctypeslib.as_array(self.L.oq_get_solution_x(self.w), shape=(self.n,)).copy()

    @property
    def y(self):
        return np.ctypeslib.as_array(self.L.oq_get_solution_y(self.w), shape=(self.m,)).copy()

    def vec(self, name, copy=True):
        ln = c_int(0)
        p = self.L.oq_get_vec(self.w, name.encode(), C.byref(ln))
        if not p:
            raise KeyError(name)
        a = np.ctypeslib.as_array(p, shape=(ln.value,))
        return a.copy() if copy else a

    def ivec(self, name):
        ln = c_int(0)
        p = self.L.oq_get_ivec(self.w, name.encode(), C.byref(ln))
        if ln.value == 0:
            return np.zeros(0, dtype=np.int64)
        return np.ctypeslib.as_array(p, shape=(ln.value,)).copy()

    def scalar(self, name):
        return float(self.L.oq_get_scalar(self.w, name.encode()))

    def set_scalar(self, name, v):
        self.L.oq_set_scalar(self.w, name.encode(), float(v))

    def set_perm(self, perm):
        """sparse-storage mode: factorise P H P' under this symmetric permutation (perm[new] = old), before the first solve"""
        a = np.ascontiguousarray(perm, dtype=np.int64)
        if self.L.oq_set_perm(self.w, a.ctypes.data_as(C.POINTER(c_int)), len(a)) != 0:
            raise ValueError("not a permutation of 0 .. n-1 (or the factor exists already)")

    def sparse_levels(self):
        return int(self.L.oq_sparse_levels(self.w))

    def counter(self, name):
        return int(self.L.oq_get_counter(self.w, name.encode()))

    def matrix(self, name):
        nr, nc = c_int(0), c_int(0)
        p, i, x = C.POINTER(c_int)(), C.POINTER(c_int)(), C.POINTER(c_float)()
        self.L.oq_get_matrix(self.w, name.encode(), C.byref(nr), C.byref(nc), C.byref(p), C.byref(i), C.byref(x))
        if nc.value == 0:
            return None
        pp = np.ctypeslib.as_array(p, shape=(nc.value + 1,)).copy()
        nz = int(pp[-1])
        ii = np.ctypeslib.as_array(i, shape=(max(nz, 1),))[:nz].copy()
        xx = np.ctypeslib.as_array(x, shape=(max(nz, 1),))[:nz].copy()
        return nr.value, nc.value, pp, ii, xx

    def factor(self):
        """Dense (L strict lower with unit diagonal implicit, D)."""
        D = C.POINTER(c_float)()
        ld = c_int(0)
        Lp = self.L.oq_get_factor(self.w, C.byref(D), C.byref(ld))
        n = self.n
        Lm = np.ctypeslib.as_array(Lp, shape=(n * n,)).reshape(n, n).T.copy()  # column-major -> [i, j]
        return np.tril(Lm, -1) + np.eye(n), np.ctypeslib.as_array(D, shape=(n,)).copy()

    def kkt_factor(self):
        """Dense factor of the (n+m) x (n+m) KKT matrix (KKT mode): (L with unit diagonal, D)."""
        D = C.POINTER(c_float)()
        ld = c_int(0)
        Lp = self.L.oq_get_kkt_factor(self.w, C.byref(D), C.byref(ld))
        nn = int(ld.value)
        Lm = np.ctypeslib.as_array(Lp, shape=(nn * nn,)).reshape(nn, nn).T.copy()
        return np.tril(Lm, -1) + np.eye(nn), np.ctypeslib.as_array(D, shape=(nn,)).copy()

    def enable_trace(self, cap):
        n, m = self.n, self.m
        bufs = dict(kind=np.zeros(cap, np.int64), fact=np.zeros(cap, np.int64), nb_active=np.zeros(cap, np.int64),
                    nb_enter=np.zeros(cap, np.int64), nb_leave=np.zeros(cap, np.int64), tau=np.zeros(cap),
                    gamma=np.zeros(cap), pri_res_norm=np.zeros(cap), dua_res_norm=np.zeros(cap),
                    dua2_res_norm=np.zeros(cap), x=np.zeros((cap, n)), y=np.zeros((cap, m)), d=np.zeros((cap, n)),
                    active=np.zeros((cap, m), np.int64))
        t = Trace()
        t.cap, t.len = cap, 0
        for k, a in bufs.items():
            setattr(t, k, iptr(a) if a.dtype == np.int64 else fptr(a))
        self._trace = (t, bufs)
        self.L.oq_set_trace(self.w, C.byref(t))

    def trace(self):
        t, bufs = self._trace
        ln = int(t.len)
        return {k: a[:ln].copy() for k, a in bufs.items()}

    def cleanup(self):
        if self.w:
            self.L.oq_cleanup(self.w)
            self.w = None

    def __del__(self):
        try:
            self.cleanup()
        except Exception:
            pass
