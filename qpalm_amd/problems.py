"""Synthetic QP generators for the configurations of BASELINE.json.

The MATLAB generators of the reference are the model (simulations/randomQP.m:32-38,
examples/qpalm_mex_demo.m:4-10, simulations/randomMPC.m:24-28,87-121); data sets themselves
are not in the reference tree.  Everything is seeded (PCG64) so CPU and GPU legs see the same
inputs.
"""
from dataclasses import dataclass

import numpy as np
import scipy.sparse as sp


@dataclass
class QP:
    """minimize 1/2 x'Qx + q'x + c  s.t. bmin <= Ax <= bmax.  Q: CSC, lower triangle only."""
    n: int
    m: int
    Qp: np.ndarray
    Qi: np.ndarray
    Qx: np.ndarray
    Ap: np.ndarray
    Ai: np.ndarray
    Ax: np.ndarray
    q: np.ndarray
    bmin: np.ndarray
    bmax: np.ndarray
    c: float = 0.0

    def args(self):
        return (self.n, self.m, self.Qp, self.Qi, self.Qx, self.Ap, self.Ai, self.Ax, self.q, self.bmin, self.bmax)

    def Q_full(self):
        L = sp.csc_matrix((self.Qx, self.Qi, self.Qp), shape=(self.n, self.n))
        L = sp.tril(L)
        return (L + sp.tril(L, -1).T).tocsc()

    def A_mat(self):
        return sp.csc_matrix((self.Ax, self.Ai, self.Ap), shape=(self.m, self.n))


def _csc(M):
    M = sp.csc_matrix(M)
    M.sort_indices()
    M.sum_duplicates()
    return M.indptr.astype(np.int64), M.indices.astype(np.int64), M.data.astype(np.float64)


def random_qp(n=1000, m=2000, density_A=0.01, density_M=0.005, seed=1000):
    """cfg2 "random-1000": A = sprandn(m,n,dA); Q = (M+M')/2 + diag(rowsum|.|+1), M = sprandn(n,n,dM);
    q ~ N(0,1); bmin = -U(0,1); bmax = U(0,1)   (BASELINE.md section 3)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    A = sp.random(m, n, density=density_A, format="csc", random_state=rng, data_rvs=rng.standard_normal)
    M = sp.random(n, n, density=density_M, format="csc", random_state=rng, data_rvs=rng.standard_normal)
    S = ((M + M.T) * 0.5).tocsc()
    rowsum = np.asarray(abs(S).sum(axis=1)).ravel()
    Qf = (S + sp.diags(rowsum + 1.0)).tocsc()
    Ql = sp.tril(Qf).tocsc()
    q = rng.standard_normal(n)
    bmin = -rng.random(m)
    bmax = rng.random(m)
    Qp, Qi, Qx = _csc(Ql)
    Ap, Ai, Ax = _csc(A)
    return QP(n, m, Qp, Qi, Qx, Ap, Ai, Ax, q, bmin, bmax)


def random_mpc_qp(T=10, nx=10, nu=5, seed=0, x_init_scale=1.0, x_init=None):
    """cfg3 "mpc-160": non-condensed MPC QP modelled on simulations/randomMPC.m:24-28,87-121 with a
    fixed terminal box instead of the MPT invariant set (no MPT here).  n = (T+1)nx + T nu,
    rows: nx(T+1) dynamics equalities (incl. x_0 = x_init), then box rows on every state and input."""
    rng = np.random.Generator(np.random.PCG64(seed))
    Adyn = sp.random(nx, nx, density=1.0, random_state=rng, data_rvs=rng.standard_normal).toarray()
    Adyn = 0.5 * (Adyn + Adyn.T)
    Adyn = 1.0 * Adyn / max(1.0, np.max(np.abs(np.linalg.eigvals(Adyn)))) * 1.02
    Bdyn = rng.standard_normal((nx, nu))
    Mq = 5 * sp.random(nx, nx, density=0.5, random_state=rng, data_rvs=rng.standard_normal).toarray()
    Qs = Mq @ Mq.T
    R = 0.01 * np.eye(nu)
    xb = 10 + 2 * rng.random(nx)
    ub = 10 + 2 * rng.random(nu)
    n = (T + 1) * nx + T * nu
    blocks = [Qs] * T + [Qs] + [R] * T
    Qf = sp.block_diag(blocks, format="csc")
    # dynamics: x_{k+1} - A x_k - B u_k = 0 ; x_0 = x_init
    rows = []
    E0 = sp.hstack([sp.eye(nx), sp.csc_matrix((nx, n - nx))])
    rows.append(E0)
    for k in range(T):
        r = sp.lil_matrix((nx, n))
        r[:, k * nx:(k + 1) * nx] = -Adyn
        r[:, (k + 1) * nx:(k + 2) * nx] = np.eye(nx)
        r[:, (T + 1) * nx + k * nu:(T + 1) * nx + (k + 1) * nu] = -Bdyn
        rows.append(r.tocsc())
    Aeq = sp.vstack(rows)
    Abox = sp.eye(n, format="csc")
    A = sp.vstack([Aeq, Abox]).tocsc()
    if x_init is None:
        x_init = x_init_scale * (2 * rng.random(nx) - 1) * 2.0
    beq = np.concatenate([x_init, np.zeros(T * nx)])
    box = np.concatenate([np.tile(xb, T + 1), np.tile(ub, T)])
    bmin = np.concatenate([beq, -box])
    bmax = np.concatenate([beq, box])
    q = np.zeros(n)
    Qp, Qi, Qx = _csc(sp.tril(Qf))
    Ap, Ai, Ax = _csc(A)
    return QP(n, A.shape[0], Qp, Qi, Qx, Ap, Ai, Ax, q, bmin, bmax)


def config5_qp(n=5000, m=5000, seed=55):
    """BASELINE.json config 5 ("nonconvex random QP n = 5000"): random_qp with every fifth diagonal entry of Q lowered by 2.5 x its
    value (indefinite Hessian).  One definition for tests/test_coop.py, tools/ and the golden fixture tests/golden/config5_n5000.npz."""
    p = random_qp(n, m, seed=seed, density_A=0.002, density_M=0.001)
    Q = sp.csc_matrix((p.Qx, p.Qi, p.Qp), shape=(n, n)).tolil()
    for j in range(0, n, 5):
        Q[j, j] = Q[j, j] - 2.5 * abs(Q[j, j])
    Q = sp.csc_matrix(Q)
    Q.sort_indices()
    return QP(n, m, Q.indptr.astype(np.int64), Q.indices.astype(np.int64), Q.data.copy(), p.Ap, p.Ai, p.Ax, p.q, p.bmin, p.bmax)


def sparse_qp(n, kind="banded", seed=0, band=3, block=8, rows_per_block=4):
    """Large SPARSE convex QPs whose Schur complement Q + A'A stays sparse under the natural ordering (round 5: the sparse factor).
      banded:  Q tridiagonal-dominant with `band` sub-diagonals, A = two-variable difference rows x_i - x_{i+1} in [-1, 1] plus box rows
               on every third variable (the elimination tree is a chain: n levels);
      blocks:  block-diagonal Q with dense `block` x `block` blocks, `rows_per_block` constraints inside each block (a forest of
               n / block small trees: `block` levels, thousands of columns per level);
      arrow:   banded, plus one dense last row / column of Q (every column of L gets one more entry)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    rows, cols, vals = [], [], []
    if kind in ("banded", "arrow"):
        for k in range(1, band + 1):
            v = 0.3 * rng.standard_normal(n - k) / k
            rows += list(range(k, n)); cols += list(range(0, n - k)); vals += list(v)
        if kind == "arrow":
            v = 0.05 * rng.standard_normal(n - band - 1)
            rows += [n - 1] * (n - band - 1); cols += list(range(0, n - band - 1)); vals += list(v)
        Lo = sp.csc_matrix((vals, (rows, cols)), shape=(n, n))
        S = Lo + Lo.T
        Qf = (S + sp.diags(np.asarray(abs(S).sum(axis=1)).ravel() + 1.0)).tocsc()
        ar, ac, av = [], [], []
        m = 0
        for i in range(0, n - 1, 2):            # difference rows
            ar += [m, m]; ac += [i, i + 1]; av += [1.0, -1.0]; m += 1
        for i in range(0, n, 3):                # box rows
            ar.append(m); ac.append(i); av.append(1.0); m += 1
        A = sp.csc_matrix((av, (ar, ac)), shape=(m, n))
        bmin, bmax = -0.5 * rng.random(m), 0.5 * rng.random(m)
    elif kind == "blocks":
        nb = n // block
        n = nb * block
        blocks = []
        ar, ac, av = [], [], []
        m = 0
        for b in range(nb):
            M = rng.standard_normal((block, block)) * 0.3
            Sb = 0.5 * (M + M.T)
            blocks.append(Sb + np.diag(np.abs(Sb).sum(axis=1) + 1.0))
            for r in range(rows_per_block):
                idx = rng.choice(block, size=min(3, block), replace=False)
                for i in idx:
                    ar.append(m); ac.append(b * block + int(i)); av.append(float(rng.standard_normal()))
                m += 1
        Qf = sp.block_diag(blocks, format="csc")
        A = sp.csc_matrix((av, (ar, ac)), shape=(m, n))
        bmin, bmax = -rng.random(m), rng.random(m)
    else:
        raise ValueError(kind)
    A.sum_duplicates(); A.sort_indices()
    Ql = sp.tril(Qf).tocsc(); Ql.sort_indices()
    q = rng.standard_normal(n)
    Qp, Qi, Qx = _csc(Ql)
    Ap, Ai, Ax = _csc(A)
    return QP(n, A.shape[0], Qp, Qi, Qx, Ap, Ai, Ax, q, bmin, bmax)


def replicated_qp(base, copies, seed=0, pert=0.05):
    """Block-diagonal QP made of `copies` randomly perturbed copies of `base` (a small fixture QP): a larger instance
    that keeps the base problem's behaviour (e.g. the reference's basic_qp reaches boost_gamma, iteration.c:158-211)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    Q, A = base.Q_full(), base.A_mat()
    Qs, As, qs = [], [], []
    for _ in range(copies):
        Qs.append(Q * (1 + pert * rng.standard_normal()))
        As.append(sp.csc_matrix(A.multiply(1 + pert * rng.standard_normal(A.shape))))
        qs.append(base.q * (1 + pert * rng.standard_normal(base.n)))
    Qf, Af = sp.block_diag(Qs, format="csc"), sp.block_diag(As, format="csc")
    Qp, Qi, Qx = _csc(sp.tril(Qf))
    Ap, Ai, Ax = _csc(Af)
    return QP(Qf.shape[0], Af.shape[0], Qp, Qi, Qx, Ap, Ai, Ax, np.concatenate(qs), np.tile(base.bmin, copies), np.tile(base.bmax, copies))


def fixture_qp(p):
    """QP from a tests/golden/reference_tests.json problem dict."""
    return QP(p["n"], p["m"], np.array(p["Qp"], np.int64), np.array(p["Qi"], np.int64), np.array(p["Qx"], float),
              np.array(p["Ap"], np.int64), np.array(p["Ai"], np.int64), np.array(p["Ax"], float),
              np.array(p["q"], float), np.array(p["bmin"], float), np.array(p["bmax"], float), float(p.get("c", 0.0)))
