"""Host-side mirror of the reference's Python interface on top of the gfx950 C ABI.

`Qpalm` mirrors interfaces/python/qpalm.py:192-375 (set_default_settings / set_data / _allocate_work ==
qpalm_setup / _solve / _warm_start / _update_bounds / _update_q) for ONE QP; `QpalmBatch` is the
batched form the MI355X engine is built for (B independent QPs of equal dimensions resident in HBM).
Neither class contains numerics: everything runs in the HIP kernels behind include/qpalm_gfx950.h.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import Info, QpgError, Settings, Stats, f64, fptr, i64, iptr

SOLVED, DUAL_TERMINATED, MAX_ITER_REACHED = 1, 2, -2
PRIMAL_INFEASIBLE, DUAL_INFEASIBLE, TIME_LIMIT_REACHED, UNSOLVED, ERROR = -3, -4, -5, -10, 0


class Context:
    def __init__(self, device=0, lib_path=None, **options):
        self.L = capi.load(lib_path)
        h = C.c_void_p()
        rc = self.L.qpg_ctx_create(int(device), C.byref(h))
        if rc != 0:
            raise QpgError(rc, self.L.qpg_last_error().decode())
        self.h = h
        for k, v in options.items():
            self.set_option(k, v)

    def set_option(self, name, value):
        rc = self.L.qpg_ctx_set_option(self.h, name.encode(), int(value))
        if rc != 0:
            raise QpgError(rc, self.L.qpg_last_error().decode())

    @property
    def backend(self):
        return self.L.qpg_backend_name().decode()

    def default_settings(self, **kw):
        s = Settings()
        self.L.qpg_set_default_settings(C.byref(s))
        for k, v in kw.items():
            if not hasattr(s, k):
                raise AttributeError(k)
            setattr(s, k, v)
        return s

    def hbm_copy_gbs(self, nbytes=1 << 30, reps=5):
        """measured copy bandwidth of the device (GB/s, read + write): the attainable-HBM yardstick of bench.py"""
        g = C.c_float(0.0)
        rc = self.L.qpg_ctx_hbm_copy_gbs(self.h, int(nbytes), int(reps), C.byref(g))
        if rc != 0:
            raise QpgError(rc, self.L.qpg_last_error().decode())
        return float(g.value)

    def pinned_array(self, shape):
        """float64 array of zeros in page-locked host memory (qpg_host_alloc): hand bounds over / take solutions in such
        arrays and the copies are DMA transfers.  The memory lives as long as the array (and this context)."""
        shape = tuple(int(s) for s in (shape if isinstance(shape, (tuple, list)) else (shape,)))
        count = int(np.prod(shape)) if shape else 1
        p = C.c_void_p()
        rc = self.L.qpg_host_alloc(self.h, max(count, 1) * 8, C.byref(p))
        if rc != 0:
            raise QpgError(rc, self.L.qpg_last_error().decode())
        buf = (C.c_double * max(count, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=np.float64, count=count).reshape(shape)
        ctx_l, ctx_h, addr = self.L, self.h, p.value
        import weakref
        weakref.finalize(buf, lambda: ctx_l.qpg_host_free(ctx_h, C.c_void_p(addr)))
        return arr

    def hbm_read_gbs(self, nbytes=1 << 30, reps=5):
        """measured read-only streaming bandwidth of the device (GB/s)"""
        g = C.c_float(0.0)
        rc = self.L.qpg_ctx_hbm_read_gbs(self.h, int(nbytes), int(reps), C.byref(g))
        if rc != 0:
            raise QpgError(rc, self.L.qpg_last_error().decode())
        return float(g.value)

    def close(self):
        if self.h:
            self.L.qpg_ctx_destroy(self.h)
            self.h = None


class QpalmBatch:
    """B QPs  min 1/2 x'Qx + q'x + c  s.t. bmin <= Ax <= bmax  with common (n, m)."""

    def __init__(self, ctx, problems, settings=None):
        """problems: sequence of objects with .args() -> (n, m, Qp, Qi, Qx, Ap, Ai, Ax, q, bmin, bmax) and .c"""
        self.ctx, self.L = ctx, ctx.L
        self.B = len(problems)
        self.n, self.m = max(int(p.n) for p in problems), max(int(p.m) for p in problems)
        self.dims = [(int(p.n), int(p.m)) for p in problems]   # members may be smaller than the batch (mixed sizes)
        nnzA = max(int(p.Ap[-1]) for p in problems)
        nnzQ = max(int(p.Qp[-1]) for p in problems)
        self.settings = settings if settings is not None else ctx.default_settings()
        h = C.c_void_p()
        import time
        t0 = time.perf_counter()
        self._check(self.L.qpg_batch_create(ctx.h, self.B, self.n, self.m, nnzA, nnzQ, C.byref(self.settings), C.byref(h)))
        self.h = h
        # every member in ONE call (qpg_batch_set_problems: arrays of pointers, the conversion to the device layout runs on host threads)
        B = self.B
        keep = []   # contiguous int64 / float64 views stay alive until the call returns
        cols = [(C.c_void_p * B)() for _ in range(9)]
        for b, p in enumerate(problems):
            arrs = (i64(p.Qp), i64(p.Qi), f64(p.Qx), i64(p.Ap), i64(p.Ai), f64(p.Ax), f64(p.q), f64(p.bmin), f64(p.bmax))
            keep.append(arrs)
            for k, a in enumerate(arrs):
                cols[k][b] = a.ctypes.data
        ns, ms = i64([int(p.n) for p in problems]), i64([int(p.m) for p in problems])
        cs = f64([float(getattr(p, "c", 0.0)) for p in problems])
        self._check(self.L.qpg_batch_set_problems(self.h, 0, B, iptr(ns), iptr(ms), cols[0], cols[1], cols[2], cols[3], cols[4], cols[5], cols[6],
                                                   fptr(cs), cols[7], cols[8]))
        del keep
        t1 = time.perf_counter()
        self._check(self.L.qpg_batch_setup(self.h))
        # what qpalm_setup does (src/qpalm.c:73-319), split as this engine does it: host-side copies / format conversion of every
        # member (qpg_batch_set_problem), then packing + upload + the device part (Ruiz scaling, derived copies; qpg_batch_setup)
        self.set_problem_s, self.batch_setup_s = t1 - t0, time.perf_counter() - t1

    def _check(self, rc):
        if rc != 0:
            raise QpgError(rc, self.L.qpg_last_error().decode())

    # -- qpalm.h API ---------------------------------------------------------------------------
    def warm_start(self, x=None, y=None):
        xs = f64(x).reshape(self.B, self.n) if x is not None else None
        ys = f64(y).reshape(self.B, self.m) if y is not None else None
        self._check(self.L.qpg_batch_warm_start(self.h, fptr(xs) if xs is not None else None, fptr(ys) if ys is not None else None))

    def warm_start_last(self):
        """qpalm_warm_start of every QP with its own last solution, device-resident (the MPC receding-horizon step)."""
        self._check(self.L.qpg_batch_warm_start_last(self.h))

    def solve(self):
        self._check(self.L.qpg_batch_solve(self.h))

    def iterate(self, k=1):
        self._check(self.L.qpg_batch_iterate(self.h, int(k)))

    def last_solve_ms(self):
        ms = C.c_float(0.0)
        self._check(self.L.qpg_batch_last_solve_ms(self.h, C.byref(ms)))
        return float(ms.value)

    def launch_shape(self):
        """(concurrent workgroups, threads per workgroup, LDS bytes per workgroup) of this batch."""
        w, t, l = capi.c_int(0), capi.c_int(0), capi.c_int(0)
        self._check(self.L.qpg_batch_launch_shape(self.h, C.byref(w), C.byref(t), C.byref(l)))
        return int(w.value), int(t.value), int(l.value)

    def sparse_info(self, b=0):
        """(nnz(L) of member b's sparse factor, bytes of the device block of all sparse factors); raises on a batch with dense factors"""
        z, d = capi.c_int(0), capi.c_int(0)
        self._check(self.L.qpg_batch_sparse_info(self.h, int(b), C.byref(z), C.byref(d)))
        return int(z.value), int(d.value)

    def sparse_perm(self, b=0):
        """(perm, levels): the ordering of member b's sparse factor (perm[new] = old; identity = natural) and the height of its elimination tree"""
        n = self.dims[b][0]
        perm = np.zeros(n, dtype=np.int64)
        lev = capi.c_int(0)
        self._check(self.L.qpg_batch_sparse_perm(self.h, int(b), perm.ctypes.data_as(C.POINTER(capi.c_int)), C.byref(lev)))
        return perm, int(lev.value)

    def num_unfinished(self):
        c = capi.c_int(0)
        self._check(self.L.qpg_batch_num_unfinished(self.h, C.byref(c)))
        return int(c.value)

    def update_settings(self, s):
        rc = self.L.qpg_batch_update_settings(self.h, C.byref(s))
        if rc == 0:
            self.settings = s
        return rc

    def update_bounds(self, bmin=None, bmax=None):
        a = f64(bmin).reshape(self.B, self.m) if bmin is not None else None
        b = f64(bmax).reshape(self.B, self.m) if bmax is not None else None
        return self.L.qpg_batch_update_bounds(self.h, fptr(a) if a is not None else None, fptr(b) if b is not None else None)

    def update_q(self, q):
        q = f64(q).reshape(self.B, self.n)
        self._check(self.L.qpg_batch_update_q(self.h, fptr(q)))

    # -- results --------------------------------------------------------------------------------
    def info(self, b=0):
        out = Info()
        self._check(self.L.qpg_batch_get_info(self.h, int(b), C.byref(out)))
        return out

    def stats(self, b=0):
        out = Stats()
        self._check(self.L.qpg_batch_get_stats(self.h, int(b), C.byref(out)))
        return out

    def solution(self, out=None):
        """(x [B][n], y [B][m]); out = (x, y): C-contiguous float64 arrays to fill (e.g. Context.pinned_array: no page
        faults, DMA at PCIe rate) instead of fresh ones"""
        if out is None:
            x, y = np.zeros((self.B, self.n)), np.zeros((self.B, self.m))
        else:
            x, y = out
            assert x.shape == (self.B, self.n) and y.shape == (self.B, self.m) and x.dtype == np.float64 and y.dtype == np.float64
            assert x.flags.c_contiguous and y.flags.c_contiguous
        self._check(self.L.qpg_batch_get_solution(self.h, fptr(x), fptr(y)))
        return x, y

    def infos(self):
        """QPALMInfo of every QP (one device-to-host copy)."""
        out = (Info * self.B)()
        self._check(self.L.qpg_batch_get_info_all(self.h, out))
        return out

    def stats_all(self):
        out = (Stats * self.B)()
        self._check(self.L.qpg_batch_get_stats_all(self.h, out))
        return out

    def begin_solve(self):
        self._check(self.L.qpg_batch_begin_solve(self.h))

    def solution_of(self, k):
        """(x, y) of member k without the padding of a mixed-size batch"""
        x, y = self.solution()
        n, m = self.dims[k]
        return x[k, :n], y[k, :m]

    def statuses(self):
        return np.array([int(i.status_val) for i in self.infos()])

    _LEN_N = {"x", "Qx", "Aty", "x_prev", "x0", "Atyh", "df", "dphi", "dphi_prev", "d", "Qd", "delta_x", "temp_n", "D", "Dinv",
              "solution_x", "q"}

    def _veclen(self, name):
        if name in self._LEN_N:
            return self.n
        if name in ("delta", "alpha"):
            return 2 * self.m
        return self.m

    def vec(self, name, b=0):
        out = np.zeros(self._veclen(name))
        self._check(self.L.qpg_batch_get_vector(self.h, name.encode(), int(b), fptr(out), len(out)))
        return out

    def named_vec(self, name, length, b=0):
        """first `length` entries of a per-QP fp64 device array by its arena name (e.g. "At_sqrt_sigma")"""
        out = np.zeros(int(length))
        self._check(self.L.qpg_batch_get_vector(self.h, name.encode(), int(b), fptr(out), len(out)))
        return out

    def set_vec(self, name, v, b=0):
        v = f64(v)
        self._check(self.L.qpg_batch_set_vector(self.h, name.encode(), int(b), fptr(v), len(v)))

    def ivec(self, name, b=0, length=None):
        out = np.zeros(self.m if length is None else int(length), np.int64)
        if len(out):
            self._check(self.L.qpg_batch_get_ivector(self.h, name.encode(), int(b), iptr(out), len(out)))
        return out

    def set_ivec(self, name, v, b=0):
        v = i64(v)
        if len(v):
            self._check(self.L.qpg_batch_set_ivector(self.h, name.encode(), int(b), iptr(v), len(v)))

    def set_scalar(self, name, v, b=0):
        self._check(self.L.qpg_batch_set_scalar(self.h, name.encode(), int(b), float(v)))

    def factor(self, b=0):
        Lm, D = np.zeros((self.n, self.n)), np.zeros(self.n)
        self._check(self.L.qpg_batch_get_factor(self.h, int(b), fptr(Lm), fptr(D), self.n))
        return Lm.T.copy(), D  # column-major buffer -> [i, j]

    def factor_rows(self, rows, b=0):
        """(L, D) of a factor slot with `rows` rows: rows = n (Schur) or n + m (KKT mode)"""
        Lm, D = np.zeros((rows, rows)), np.zeros(rows)
        self._check(self.L.qpg_batch_get_factor(self.h, int(b), fptr(Lm), fptr(D), int(rows)))
        return Lm.T.copy(), D

    # -- solver_interface.h surface ---------------------------------------------------------------
    def mat_vec(self, which, x, b=0):
        x = f64(x)
        y = np.zeros(self.m if which == "A" else self.n)
        self._check(self.L.qpg_mat_vec(self.h, int(b), ord(which), fptr(x), fptr(y)))
        return y

    def mat_tpose_vec(self, which, x, b=0):
        x = f64(x)
        y = np.zeros(self.n)
        self._check(self.L.qpg_mat_tpose_vec(self.h, int(b), ord(which), fptr(x), fptr(y)))
        return y

    def op(self, name, b=0):
        self._check(getattr(self.L, "qpg_" + name)(self.h, int(b)))

    def exact_linesearch(self, b=0):
        t = capi.c_float(0.0)
        self._check(self.L.qpg_exact_linesearch(self.h, int(b), C.byref(t)))
        return float(t.value)

    def ldlsolve_all(self, reps=1):
        ms = C.c_float(0.0)
        self._check(self.L.qpg_batch_ldlsolve_all(self.h, int(reps), C.byref(ms)))
        return float(ms.value)

    def sweep_probe(self, reps=1, nranks=16):
        """diagnostic: the update sweep alone on every resident workgroup (qpg_batch_sweep_probe); ms of the launch"""
        ms = C.c_float(0.0)
        self._check(self.L.qpg_batch_sweep_probe(self.h, int(reps), int(nranks), C.byref(ms)))
        return float(ms.value)

    def device_ptr(self, name):
        p, nbytes = C.c_void_p(), C.c_size_t(0)
        self._check(self.L.qpg_batch_device_ptr(self.h, name.encode(), C.byref(p), C.byref(nbytes)))
        return p.value, nbytes.value

    def close(self):
        if getattr(self, "h", None):
            self.L.qpg_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Qpalm:
    """Single-QP facade with the reference's Python class shape (interfaces/python/qpalm.py:192-375)."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.settings = ctx.default_settings()
        self._prob = None
        self._batch = None

    def set_default_settings(self):
        self.settings = self.ctx.default_settings()

    def set_data(self, Q, A, q, bmin, bmax, c=0.0):
        """Q, A: scipy sparse (any format); only tril(Q) is read, like stype = -1."""
        import scipy.sparse as sp
        from .problems import QP
        Qc = sp.csc_matrix(sp.tril(sp.csc_matrix(Q)))
        Ac = sp.csc_matrix(A)
        Qc.sort_indices()
        Ac.sort_indices()
        self._prob = QP(Ac.shape[1], Ac.shape[0], Qc.indptr.astype(np.int64), Qc.indices.astype(np.int64),
                        Qc.data.astype(float), Ac.indptr.astype(np.int64), Ac.indices.astype(np.int64),
                        Ac.data.astype(float), f64(q), f64(bmin), f64(bmax), float(c))

    def set_problem(self, prob):
        self._prob = prob

    def setup(self):
        self._batch = QpalmBatch(self.ctx, [self._prob], self.settings)
        return self

    def warm_start(self, x=None, y=None):
        self._batch.warm_start(None if x is None else f64(x)[None, :], None if y is None else f64(y)[None, :])

    def solve(self):
        self._batch.solve()
        return self.info

    def update_settings(self, s):
        return self._batch.update_settings(s)

    def update_bounds(self, bmin=None, bmax=None):
        return self._batch.update_bounds(None if bmin is None else f64(bmin)[None, :], None if bmax is None else f64(bmax)[None, :])

    def update_q(self, q):
        self._batch.update_q(f64(q)[None, :])

    @property
    def info(self):
        return self._batch.info(0)

    @property
    def status_val(self):
        return int(self.info.status_val)

    @property
    def x(self):
        return self._batch.solution()[0][0]

    @property
    def y(self):
        return self._batch.solution()[1][0]

    @property
    def batch(self):
        return self._batch
