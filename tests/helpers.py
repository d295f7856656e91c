"""Shared helpers for the test-suite (problem construction from the golden fixtures)."""
import numpy as np

STATUS = {"SOLVED": 1, "DUAL_TERMINATED": 2, "MAX_ITER_REACHED": -2, "PRIMAL_INFEASIBLE": -3,
          "DUAL_INFEASIBLE": -4, "TIME_LIMIT_REACHED": -5, "UNSOLVED": -10, "ERROR": 0}


def prob_args(p):
    """(n, m, Qp, Qi, Qx, Ap, Ai, Ax, q, bmin, bmax) from a fixture problem dict."""
    return (p["n"], p["m"], np.array(p["Qp"], np.int64), np.array(p["Qi"], np.int64), np.array(p["Qx"], float),
            np.array(p["Ap"], np.int64), np.array(p["Ai"], np.int64), np.array(p["Ax"], float),
            np.array(p["q"], float), np.array(p["bmin"], float), np.array(p["bmax"], float))


def dense_from_csc(nrow, ncol, p, i, x, sym_lower=False):
    M = np.zeros((nrow, ncol))
    for j in range(ncol):
        for k in range(p[j], p[j + 1]):
            if sym_lower:
                if i[k] >= j:
                    M[i[k], j] = x[k]
                    M[j, i[k]] = x[k]
            else:
                M[i[k], j] += x[k]
    return M
