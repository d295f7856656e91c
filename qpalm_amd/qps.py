"""QPS front-end (BASELINE.json config 4): reads QPS files with the host C reader (include/qpalm_qps.h, the mirror of
interfaces/qps/src/qpalm_qps.c) and streams them to the GPU as size-bucketed batches (qpg_batch_set_problem_sized).
No numerics here: parsing is host C, solving is the HIP library."""
import ctypes as C
import os

import numpy as np

from .problems import QP

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB = os.path.join(_HERE, "lib", "libqpalm.so")


class _Sparse(C.Structure):
    _fields_ = [("nrow", C.c_size_t), ("ncol", C.c_size_t), ("nzmax", C.c_size_t), ("p", C.c_void_p), ("i", C.c_void_p), ("nz", C.c_void_p),
                ("x", C.c_void_p), ("z", C.c_void_p), ("stype", C.c_int), ("itype", C.c_int), ("xtype", C.c_int), ("dtype", C.c_int),
                ("sorted", C.c_int), ("packed", C.c_int)]


class _Data(C.Structure):
    _fields_ = [("n", C.c_size_t), ("m", C.c_size_t), ("Q", C.POINTER(_Sparse)), ("A", C.POINTER(_Sparse)), ("q", C.POINTER(C.c_double)),
                ("c", C.c_double), ("bmin", C.POINTER(C.c_double)), ("bmax", C.POINTER(C.c_double))]


_LIB = {}


def _lib(path=None):
    path = path or HOST_LIB
    if path not in _LIB:
        if not os.path.exists(path):
            raise ImportError("%s is missing: build first (python -c 'import __graft_entry__ as g; g.build()')" % path)
        L = C.CDLL(path)
        L.qpalm_qps_read.argtypes = [C.c_char_p, C.POINTER(C.POINTER(_Data)), C.c_char_p, C.c_size_t]
        L.qpalm_qps_free_data.argtypes = [C.POINTER(_Data)]
        L.qpalm_qps_free_data.restype = None
        _LIB[path] = L
    return _LIB[path]


def _arr(ptr, count, ctype, dtype):
    if not count:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), shape=(count,)).astype(dtype).copy()


def read_qps(path, host_lib=None):
    """QPS file -> QP (A with the variable-bound rows appended, as the reference's reader builds it)."""
    L = _lib(host_lib)
    d = C.POINTER(_Data)()
    err = C.create_string_buffer(256)
    if L.qpalm_qps_read(os.fsencode(path), C.byref(d), err, len(err)) != 0:
        raise ValueError(err.value.decode())
    try:
        D = d.contents
        n, m = int(D.n), int(D.m)
        A, Q = D.A.contents, D.Q.contents
        Ap = _arr(A.p, n + 1, C.c_int64, np.int64)
        Qp = _arr(Q.p, n + 1, C.c_int64, np.int64)
        return QP(n, m, Qp, _arr(Q.i, int(Qp[-1]), C.c_int64, np.int64), _arr(Q.x, int(Qp[-1]), C.c_double, np.float64),
                  Ap, _arr(A.i, int(Ap[-1]), C.c_int64, np.int64), _arr(A.x, int(Ap[-1]), C.c_double, np.float64),
                  _arr(D.q, n, C.c_double, np.float64), _arr(D.bmin, m, C.c_double, np.float64), _arr(D.bmax, m, C.c_double, np.float64),
                  float(D.c))
    finally:
        L.qpalm_qps_free_data(d)


def bucket_by_size(problems, max_waste=0.5):
    """Greedy size buckets for mixed-size batches: problems sorted by (n, m); a bucket grows while its smallest member
    still uses at least (1 - max_waste) of the bucket's n and m.  Returns lists of indices."""
    order = sorted(range(len(problems)), key=lambda k: (problems[k].n, problems[k].m))
    buckets, cur = [], []
    for k in order:
        if cur:
            n0, m0 = problems[cur[0]].n, max(problems[j].m for j in cur)
            n1, m1 = problems[k].n, max(m0, problems[k].m)
            if n0 < (1 - max_waste) * n1 or min(problems[j].m for j in cur + [k]) < (1 - max_waste) * m1:
                buckets.append(cur)
                cur = []
        cur.append(k)
    if cur:
        buckets.append(cur)
    return buckets


def solve_qps_files(ctx, paths, settings=None, rank=0, world=1, host_lib=None, dist=None, device=None):
    """BASELINE.json config 4: streams QPS files (free or old fixed format) as size-bucketed batches.
    Without `dist`: this process is rank `rank` of `world` and takes its share of the files, sorted by size first so that
    the shards are balanced (size-sorted round robin); returns {path: (x, y, info)} for its files.
    With `dist` (a torch.distributed module whose process group is initialised: nccl = RCCL on the GPUs, gloo in the CPU
    tests): the same sharding, one gather of x, y and the QPALMInfo records to rank 0, which returns
    {path: (x, y, info_row)} for ALL files (info_row as qpalm_amd.dist.INFO_FIELDS); the other ranks return None."""
    from .solver import QpalmBatch
    probs = [read_qps(p, host_lib) for p in paths]
    if dist is not None:
        from .dist import solve_sharded
        res = solve_sharded(probs, lambda ps: QpalmBatch(ctx, ps, settings), dist=dist, device=device)
        if res is None:
            return None
        X, Y, I = res
        return {paths[k]: (X[k, :probs[k].n].copy(), Y[k, :probs[k].m].copy(), I[k].copy()) for k in range(len(paths))}
    from .dist import shard_indices
    mine = shard_indices(len(paths), world, rank, [(p.n, p.m) for p in probs])
    out = {}
    sub = [probs[k] for k in mine]
    for bucket in bucket_by_size(sub):
        bt = QpalmBatch(ctx, [sub[k] for k in bucket], settings)
        bt.solve()
        infos = bt.infos()
        X, Y = bt.solution()   # one copy per bucket
        for pos, k in enumerate(bucket):
            nk, mk = bt.dims[pos]
            out[paths[mine[k]]] = (X[pos, :nk].copy(), Y[pos, :mk].copy(), infos[pos])
        bt.close()
    return out
