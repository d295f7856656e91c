"""Golden fixture of BASELINE.json config 5 (nonconvex random QP, n = 5000) -- BUILD CONTAINER ONLY.

Runs the CPU oracle (oracle/qpalm_oracle.c, the standard -O3 -ffp-contract=off build) ONCE on qpalm_amd.problems.config5_qp() with the
settings of tests/test_coop.py::test_config5_nonconvex_n5000 and writes x, y, the status, the iteration counts and the refactor /
rank-update counts to tests/golden/config5_n5000.npz (80 KB of solution data).  The oracle's dense n = 5000 factorisations make this
a run of one to two hours on one core; the GPU test compares the engine with the file, it never runs this script.

    python tests/golden/make_config5_fixture.py [--eps 1e-6]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle.binding as ob                      # noqa: E402
from qpalm_amd.problems import config5_qp        # noqa: E402

if __name__ == "__main__":
    eps = float(sys.argv[sys.argv.index("--eps") + 1]) if "--eps" in sys.argv else 1e-6
    st = dict(eps_abs=eps, eps_rel=eps, verbose=0, nonconvex=1, max_iter=40000)
    p = config5_qp()
    t0 = time.time()
    o = ob.OracleQP(*p.args(), settings=ob.default_settings(**st))
    t1 = time.time()
    print("setup %.1f s, lobpcg lambda %.12g, %d lobpcg iterations" % (t1 - t0, o.scalar("lobpcg_lambda"), o.counter("n_lobpcg_iter")), flush=True)
    o.solve()
    t2 = time.time()
    out = dict(x=o.x, y=o.y, status_val=np.int64(o.status_val), iter=np.int64(o.info.iter), iter_out=np.int64(o.info.iter_out),
               objective=np.float64(o.info.objective), pri_res_norm=np.float64(o.info.pri_res_norm), dua_res_norm=np.float64(o.info.dua_res_norm),
               lobpcg_lambda=np.float64(o.scalar("lobpcg_lambda")), n_lobpcg_iter=np.int64(o.counter("n_lobpcg_iter")),
               n_refactor=np.int64(o.counter("n_refactor")), n_factor_Q=np.int64(o.counter("n_factor_Q")), n_rank1=np.int64(o.counter("n_rank1")),
               n_solve=np.int64(o.counter("n_solve")), eps=np.float64(eps), oracle_solve_seconds=np.float64(t2 - t1))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "config5_n5000.npz"), **out)
    print("solve %.1f s: status %d, %d iterations (%d outer), %d refactorisations, %d rank-1 updates" % (
        t2 - t1, int(o.status_val), int(o.info.iter), int(o.info.iter_out), o.counter("n_refactor"), o.counter("n_rank1")), flush=True)
