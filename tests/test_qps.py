"""QPS front-end (BASELINE.json config 4): the host C reader (include/qpalm_qps.h, mirror of
interfaces/qps/src/qpalm_qps.c:71-537,610-689) on hand-written fixtures under tests/golden/qps/, an independent pure-Python
parse of the same files as the checker of the parse, and the files streamed through size-bucketed (mixed-size) batches.
The Maros-Meszaros set itself is not in the reference tree and there is no network."""
import ctypes as C
import glob
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import binding as ob
from qpalm_amd import build
from qpalm_amd.problems import random_qp
from qpalm_amd.qps import bucket_by_size, read_qps
from qpalm_amd.solver import QpalmBatch
from tests.test_parity import RTOL, rel

QDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "qps")
INF = 1e20


@pytest.fixture(scope="module")
def host_lib(emu_lib):
    return build.build_host_emu()      # the reader is host C; the emulated backend only satisfies the linker here


def py_parse(path):
    """independent, dictionary-based reading of a free-format QPS file with the reference's conventions (the checker)"""
    sec, rows, obj, cols, colnames = None, {}, None, {}, []
    rhs, ranges, bounds, quad, c = {}, {}, [], [], 0.0
    for ln in open(path):
        if ln.startswith("*") or not ln.strip():
            continue
        if not ln[0].isspace():
            sec = ln.split()[0]
            continue
        t = ln.split()
        if sec == "ROWS":
            if t[0] == "N":
                obj = obj or t[1]
            else:
                rows[t[1]] = (len(rows), t[0])
        elif sec == "COLUMNS":
            if t[0] not in cols:
                cols[t[0]] = {}
                colnames.append(t[0])
            for k in range(1, len(t) - 1, 2):
                cols[t[0]][t[k]] = float(t[k + 1])
        elif sec in ("RHS", "RANGES"):
            k0 = 1 if len(t) % 2 else 0
            for k in range(k0, len(t) - 1, 2):
                (rhs if sec == "RHS" else ranges)[t[k]] = float(t[k + 1])
        elif sec == "BOUNDS":
            if t[0] in ("FR", "MI", "PL"):
                bounds.append((t[0], t[-1], None))
            else:
                bounds.append((t[0], t[-2], float(t[-1])))
        elif sec == "QUADOBJ":
            quad.append((t[0], t[1], float(t[2])))
    n, mr = len(colnames), len(rows)
    free = {b[1] for b in bounds if b[0] == "FR"}
    brow, nxt = {}, mr
    for name in colnames:
        if name not in free:
            brow[name] = nxt
            nxt += 1
    m = nxt
    A = np.zeros((m, n)); q = np.zeros(n); bmin = np.zeros(m); bmax = np.zeros(m)
    for name, (r, sgn) in rows.items():
        bmin[r], bmax[r] = {"L": (-INF, 0.0), "G": (0.0, INF), "E": (0.0, 0.0)}[sgn]
    for j, name in enumerate(colnames):
        for rname, v in cols[name].items():
            if rname == obj:
                q[j] = v
            else:
                A[rows[rname][0], j] = v
        if name in brow:
            A[brow[name], j] = 1.0
            bmin[brow[name]], bmax[brow[name]] = 0.0, INF
    for rname, v in rhs.items():
        if rname == obj:
            c = -v
            continue
        r, sgn = rows[rname]
        if sgn == "L": bmax[r] = v
        elif sgn == "G": bmin[r] = v
        else: bmin[r] = bmax[r] = v
    for rname, v in ranges.items():
        r, sgn = rows[rname]
        if sgn == "L": bmin[r] = bmax[r] - v
        elif sgn == "G": bmax[r] = bmin[r] + v
        elif v >= 0: bmax[r] = bmin[r] + v
        else: bmin[r] = bmax[r] + v
    for typ, name, v in bounds:
        if typ == "FR":
            continue
        r = brow[name]
        if typ == "UP": bmax[r] = v
        elif typ == "LO": bmin[r] = v
        elif typ == "FX": bmin[r] = bmax[r] = v
        elif typ == "MI": bmin[r] = -INF
        elif typ == "PL": bmax[r] = INF
    Ql = np.zeros((n, n))
    idx = {name: j for j, name in enumerate(colnames)}
    for c1, c2, v in quad:
        Ql[idx[c2], idx[c1]] = v       # column c1, row c2: lower triangle, column major
    return n, m, Ql, A, q, c, bmin, bmax


@pytest.mark.parametrize("name", ["qptest", "ranges_free", "chain6"])
def test_reader_matches_independent_parse(host_lib, name):
    path = os.path.join(QDIR, name + ".qps")
    p = read_qps(path, host_lib)
    n, m, Ql, A, q, c, bmin, bmax = py_parse(path)
    assert (p.n, p.m) == (n, m) and p.c == c
    assert np.array_equal(sp.csc_matrix((p.Ax, p.Ai, p.Ap), shape=(m, n)).toarray(), A)
    assert np.array_equal(sp.csc_matrix((p.Qx, p.Qi, p.Qp), shape=(n, n)).toarray(), Ql)
    assert np.array_equal(p.q, q) and np.array_equal(p.bmin, bmin) and np.array_equal(p.bmax, bmax)
    # the identity (bound) entry comes first in its column, before the regular entries (qpalm_qps.c:316-324)
    for j in range(n):
        col = p.Ai[p.Ap[j]:p.Ap[j + 1]]
        if len(col) and np.any(col >= m - (n - (name == "ranges_free"))):
            assert col[0] == col.max()


def test_reader_errors_and_settings_file(host_lib, tmp_path):
    with pytest.raises(ValueError, match="Could not open"):
        read_qps(str(tmp_path / "missing.qps"), host_lib)
    bad = tmp_path / "bad.qps"
    bad.write_text("NAME X\nROWS\n N obj\n L r1\nCOLUMNS\n    c1 r9 1.0\nENDATA\n")
    with pytest.raises(ValueError, match="Unknown row"):
        read_qps(str(bad), host_lib)
    twice = tmp_path / "twice.qps"
    twice.write_text("NAME X\nROWS\n N obj\n L r1\nCOLUMNS\n    c1 r1 1.0\n    c2 r1 1.0\n    c1 obj 2.0\nENDATA\n")
    with pytest.raises(ValueError, match="appears in two places"):
        read_qps(str(twice), host_lib)
    intb = tmp_path / "intb.qps"
    intb.write_text("NAME X\nROWS\n N obj\n L r1\nCOLUMNS\n    c1 r1 1.0\nBOUNDS\n BV BND c1\nENDATA\n")
    with pytest.raises(ValueError, match="Malformed BOUNDS|Unsupported bound type"):
        read_qps(str(intb), host_lib)
    longl = tmp_path / "long.qps"
    longl.write_text("NAME X\nROWS\n N obj\n L r1" + " " * 600 + "\nENDATA\n")
    with pytest.raises(ValueError, match="Line too long"):
        read_qps(str(longl), host_lib)
    from qpalm_amd.capi import Settings
    L = C.CDLL(host_lib)
    s = Settings()
    err = C.create_string_buffer(256)
    assert L.qpalm_qps_read_settings(os.fsencode(os.path.join(QDIR, "settings_sample.txt")), C.byref(s), err, 256) == 0
    assert (s.eps_abs, s.eps_rel, s.max_iter, s.verbose, s.scaling) == (1e-7, 1e-7, 5000, 0, 2)
    assert s.gamma_init == 1e7 and s.inner_max_iter == 100        # untouched defaults
    badset = tmp_path / "set.txt"
    badset.write_text("a\nb\nc\nd\ne\nnot_a_setting 3\n")
    assert L.qpalm_qps_read_settings(os.fsencode(str(badset)), C.byref(s), err, 256) != 0 and b"Unrecognised setting" in err.value


def _fixed_paths():
    return sorted(glob.glob(os.path.join(QDIR, "fixed", "fix*.qps")))


def test_fixed_format_reader_matches_the_generated_data(host_lib):
    """The old fixed-column format (names with blanks; the form the Maros-Meszaros files come in; the reference converts such
    files first, interfaces/qps/src/qps_conversion.c:36-146): the 50 synthetic files of tests/golden/qps/make_fixed_qps.py against
    the data the generator encoded in them, entry by entry (every number in the files is exact)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_fixed_qps", os.path.join(QDIR, "make_fixed_qps.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    paths = _fixed_paths()
    assert len(paths) == gen.COUNT == 50
    blanks = 0
    for k, path in enumerate(paths):
        P = gen.problem(k)
        E = gen.expected(P)
        p = read_qps(path, host_lib)
        assert (p.n, p.m) == (E["n"], E["m"]) and p.c == E["c"], path
        assert np.array_equal(sp.csc_matrix((p.Ax, p.Ai, p.Ap), shape=(p.m, p.n)).toarray(), E["A"]), path
        assert np.array_equal(sp.csc_matrix((p.Qx, p.Qi, p.Qp), shape=(p.n, p.n)).toarray(), np.tril(E["Q"])), path
        assert np.array_equal(p.q, E["q"]) and np.array_equal(p.bmin, E["bmin"]) and np.array_equal(p.bmax, E["bmax"]), path
        blanks += P["blank"]
    assert blanks == 40     # forty files have names with embedded blanks, ten have plain names in fixed columns
    # the generator is deterministic: the committed files are what it writes
    import io
    for k in (0, 7, 49):
        tmp = os.path.join(os.environ.get("TMPDIR", "/tmp"), "fix_regen_%d_%d.qps" % (os.getpid(), k))
        gen.write(gen.problem(k), tmp)
        assert open(tmp).read() == open(paths[k]).read()
        os.remove(tmp)


def _check_against_oracle(paths, results, host_lib, st):
    """every file of `results` {path: (x, y, info)} against the oracle on the reader's data"""
    from qpalm_amd.dist import INFO_FIELDS
    nsolved = 0
    for path in paths:
        p = read_qps(path, host_lib)
        o = ob.OracleQP(*p.args(), c=p.c, settings=ob.default_settings(**st))
        o.solve()
        x, y, info = results[path]
        if isinstance(info, np.ndarray):
            status, it, obj = int(info[INFO_FIELDS.index("status_val")]), int(info[INFO_FIELDS.index("iter")]), float(info[INFO_FIELDS.index("objective")])
        else:
            status, it, obj = int(info.status_val), int(info.iter), float(info.objective)
        assert status == o.status_val, (path, status, o.status_val)
        assert it == int(o.info.iter), (path, it, int(o.info.iter))
        if status == 1:
            nsolved += 1
            assert rel(x, o.x) <= RTOL and rel(y, o.y) <= RTOL, path
            assert abs(obj - o.info.objective) <= 1e-9 * max(1.0, abs(o.info.objective)), path
    return nsolved


def _qps_rank(rank, world, port, emu_lib, host_lib, paths, st, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from qpalm_amd.qps import solve_qps_files
    from qpalm_amd.solver import Context
    ctx = Context(0, lib_path=emu_lib)
    res = solve_qps_files(ctx, paths, ctx.default_settings(**st), host_lib=host_lib, dist=dist)
    if rank == 0:
        q.put({k: (v[0], v[1], v[2]) for k, v in res.items()})
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_fixed_qps_files_streamed_over_two_ranks(emu_lib, host_lib):
    """Config 4's entry point, solve_qps_files: fixed-format files sharded over two gloo ranks (size-sorted round robin,
    size buckets inside a shard), ONE gather to rank 0; every file against the oracle (status, iteration count, x, y, objective)."""
    import torch.multiprocessing as mp
    paths = _fixed_paths()[::3]          # 17 of the 50 in the emulator (all 50 on the GPU, test_fixed_qps_files_on_gfx950)
    st = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0)
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    port = 29500 + ((os.getpid() + 977) % 2000)
    procs = [ctxm.Process(target=_qps_rank, args=(r, 2, port, emu_lib, host_lib, paths, st, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=900)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(res) == sorted(paths)
    assert _check_against_oracle(paths, res, host_lib, st) >= 10


@pytest.mark.gpu
def test_fixed_qps_files_on_gfx950():
    """the same 50 files through solve_qps_files on the HIP library (host reader = the shipped libqpalm.so): as one rank, and
    as the two shards two ranks would take, each file against the oracle"""
    from qpalm_amd.qps import solve_qps_files
    from qpalm_amd.solver import Context
    ctx = Context(0)
    assert ctx.backend == "gfx950-hip"
    paths = _fixed_paths()
    st = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0)
    one = solve_qps_files(ctx, paths, ctx.default_settings(**st))
    assert _check_against_oracle(paths, one, None, st) >= 30
    two = {}
    for r in range(2):
        part = solve_qps_files(ctx, paths, ctx.default_settings(**st), rank=r, world=2)
        assert 24 <= len(part) <= 26
        two.update(part)
    assert sorted(two) == sorted(paths)
    for path in paths:
        assert np.array_equal(one[path][0], two[path][0]) and np.array_equal(one[path][1], two[path][1])   # batch composition does not change results


def test_qps_files_as_one_mixed_size_batch(ctx, request):
    """The fixtures (n = 2, 5, 6; m = 4, 8, 12) plus two random QPs of other sizes in ONE batch (members keep their own
    dimensions on the device), against the oracle on each file's data; QPTEST's published optimum as a known answer.
    The files are read by the host library of the backend under test: the shipped libqpalm.so on the hardware ([hip]), the
    emulator-linked build of the same C source on the CPU ([emu])."""
    reader = None if ctx.kind == "hip" else request.getfixturevalue("host_lib")
    paths = sorted(glob.glob(os.path.join(QDIR, "*.qps")))
    probs = [read_qps(p, reader) for p in paths] + [random_qp(12, 20, seed=3, density_A=0.3, density_M=0.2), random_qp(7, 30, seed=4, density_A=0.3, density_M=0.3)]
    st = dict(eps_abs=1e-7, eps_rel=1e-7, verbose=0)
    bt = QpalmBatch(ctx, probs, ctx.default_settings(**st))
    assert (bt.n, bt.m) == (12, 30)
    bt.solve()
    infos = bt.infos()
    for k, p in enumerate(probs):
        o = ob.OracleQP(*p.args(), c=p.c, settings=ob.default_settings(**st))
        o.solve()
        x, y = bt.solution_of(k)
        assert int(infos[k].status_val) == o.status_val == 1, (k, infos[k].status_val)
        assert int(infos[k].iter) == int(o.info.iter)
        assert rel(x, o.x) <= RTOL and rel(y, o.y) <= RTOL
        assert abs(infos[k].objective - o.info.objective) <= 1e-9 * max(1.0, abs(o.info.objective))
        X, Y = bt.solution()
        assert not np.any(X[k, p.n:]) and not np.any(Y[k, p.m:])      # padding stays zero
    kq = [os.path.basename(p) for p in paths].index("qptest.qps")
    assert np.max(np.abs(bt.solution_of(kq)[0] - np.array([0.7625, 0.475]))) <= 1e-5
    assert abs(infos[kq].objective - 8.371875) <= 1e-5           # 4.371875 of the Maros-Meszaros table + the constant 4
    # size buckets: members within a factor of two of the bucket's dimensions
    bk = bucket_by_size(probs)
    assert sorted(sum(bk, [])) == list(range(len(probs))) and all(probs[b[0]].n >= 0.5 * probs[b[-1]].n for b in bk)


def test_mixed_sizes_in_kkt_mode_and_through_the_queue(ctx):
    sizes_ = [(10, 18), (16, 30), (13, 22), (16, 12), (9, 30)]
    probs = [random_qp(n, m, seed=70 + k, density_A=0.3, density_M=0.2) for k, (n, m) in enumerate(sizes_)]
    ctx.set_option("max_slots", 2)
    try:
        for extra in (dict(), dict(factorization_method=0)):
            st = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0, **extra)
            bt = QpalmBatch(ctx, probs, ctx.default_settings(**st))
            bt.solve()
            for k, p in enumerate(probs):
                o = ob.OracleQP(*p.args(), settings=ob.default_settings(**st))
                o.solve()
                x, y = bt.solution_of(k)
                assert int(bt.info(k).status_val) == o.status_val == 1 and int(bt.info(k).iter) == int(o.info.iter)
                assert rel(x, o.x) <= RTOL and rel(y, o.y) <= RTOL
    finally:
        ctx.set_option("max_slots", 512)


@pytest.mark.gpu
def test_qps_cli_on_gfx950():
    """the reference's CLI shape (main, qpalm_qps.c:691-831): qpalm_qps problem.qps settings.txt on the HIP backend"""
    import subprocess
    exe = os.path.join(os.path.dirname(build.LIB), "qpalm_qps")
    r = subprocess.run([exe, os.path.join(QDIR, "qptest.qps"), os.path.join(QDIR, "settings_sample.txt")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Reading successful." in r.stdout and "Iter:" in r.stdout and "Status: solved" in r.stdout
    obj = float(r.stdout.split("objective")[1].split()[0])
    assert abs(obj - 8.371875) <= 1e-5
