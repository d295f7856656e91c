#!/usr/bin/env python3
"""bench.py -- QP solves/sec on batched QPs, BASELINE.json's metric.

Workloads (--workload):
  random-1000 (default, BASELINE.json configs[1]): n=1000, m=2000, ~1 % dense A, cold start, eps 1e-6.
  mpc-160     (configs[2]): n=160, m=270 non-condensed MPC QPs (T=10, nx=10, nu=5), perturbed initial states,
              update_bounds + warm start between steps (simulations/randomMPCsequential.m:158-177).

One "step" = one pass of the hot path over one batch: every QP of the (HBM-resident, already scaled) batch is
(re)started and solved by the persistent gfx950 kernel.  With N > 1 ranks every rank owns its own shard of B QPs
(weak scaling, no data-path collective) and solution.x, solution.y and the QPALMInfo records are gathered to rank 0
over RCCL inside the timed region.

`python bench.py --gpus N` without a torch.distributed environment starts N ranks itself (one child process per
GPU, started BEFORE this process imports torch or touches HIP); under torchrun the environment is used as is.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (k_solve): algorithmic bytes of SURVEY.md
section 8d, counted from device-side work counters, divided by the kernel's HIP-event duration; fractions are quoted
against the 8 TB/s spec and against a copy kernel measured in the same run.  `cpu_baseline` times the CPU oracle
("port") on a bounded sample of the same QPs.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def lib_sha256(path=None):
    """hash of the DEVICE CODE of the built HIP library that is loaded (a stale .so must not have another build's counters quoted for
    it): the .text and .rodata sections of the gfx950 code object inside the shared object's `.hip_fatbin` section.  Not the whole
    file -- hipcc gives every compilation a random `__hip_cuid_*` symbol, so two builds of the same sources differ in a few hundred
    bytes of the host AND device symbol tables while the machine code is identical (a rebuild on another machine -- the driver's
    build() -- must still find its counters)."""
    import hashlib
    import struct
    from qpalm_amd import capi
    with open(path or capi.LIB_PATH, "rb") as fh:
        blob = fh.read()

    def sections(elf):   # ELF64 little endian: {name: bytes}
        if elf[:4] != b"\x7fELF" or elf[4] != 2:
            raise ValueError("not ELF64")
        shoff, = struct.unpack_from("<Q", elf, 0x28)
        shentsize, shnum, shstrndx = struct.unpack_from("<HHH", elf, 0x3A)
        hdr = [struct.unpack_from("<IIQQQQ", elf, shoff + k * shentsize) for k in range(shnum)]
        names = elf[hdr[shstrndx][4]:hdr[shstrndx][4] + hdr[shstrndx][5]]
        return {names[h[0]:names.index(b"\0", h[0])]: elf[h[4]:h[4] + h[5]] for h in hdr if h[1] != 8}   # (8 = SHT_NOBITS)
    try:
        fat = sections(blob)[b".hip_fatbin"]
        if fat[:24] != b"__CLANG_OFFLOAD_BUNDLE__":
            raise ValueError("unexpected bundle format")
        count, = struct.unpack_from("<Q", fat, 24)
        pos, h, found = 32, hashlib.sha256(), False
        for _ in range(count):
            off, size, tlen = struct.unpack_from("<QQQ", fat, pos)
            triple = fat[pos + 24:pos + 24 + tlen]
            pos += 24 + tlen
            if b"gfx950" in triple and size:
                dev = sections(fat[off:off + size])
                for name in (b".text", b".rodata"):
                    h.update(name)
                    h.update(dev.get(name, b""))
                found = True
        if not found:
            raise ValueError("no gfx950 code object")
        return h.hexdigest()
    except Exception:   # noqa: BLE001 -- an unexpected layout: the whole file (never matches another build, which is the safe side)
        return hashlib.sha256(blob).hexdigest()


def source_sha256():
    """hash of the kernel + C-ABI sources: a PMC summary is only quoted for the build it was measured on"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "qpalm_amd", "csrc")
    for f in sorted(os.listdir(d)):
        with open(os.path.join(d, f), "rb") as fh:
            h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()




def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="random-1000", choices=("random-1000", "mpc-160", "sparse-banded-2000", "sparse-blocks-2000"),
                    help="random-1000: the headline (BASELINE.json configs[1]); mpc-160: config 3; sparse-*: a batch of sparse QPs with n = 2000 on the "
                         "sparse factor (SURVEY section 8 row h; default batch 2048): QP/s and the phase split, no byte model")
    ap.add_argument("--batch", type=int, default=int(os.environ.get("QPALM_BENCH_BATCH", "8192")),
                    help="QPs per GPU (512 resident factor slots = workgroups; the rest queue up behind them)")
    ap.add_argument("--n", type=int, default=0, help="random workload: number of variables (default 1000)")
    ap.add_argument("--m", type=int, default=0, help="random workload: number of constraints (default 2 n)")
    ap.add_argument("--rank-threshold", type=int, default=int(os.environ.get("QPALM_RANK_THRESHOLD", "-1")))
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-mpc", action="store_true", help="skip the short mpc-160 lines (BASELINE.json config 3) appended to the default run's JSON")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (nccl = RCCL) and run the gather of x, y and the info records even with ONE rank: "
                         "executes the multi-GPU path's RCCL initialisation and device-view gather on a single-GPU box (tests/test_bench_launcher.py)")
    ap.add_argument("--kkt", action="store_true", help="factorization_method = FACTORIZE_KKT: the (n+m) x (n+m) KKT panel with row additions / deletions "
                                                        "(what BASELINE.json config 3 literally names) instead of the Schur panel with rank updates")
    ap.add_argument("--small-workgroups", type=int, default=1, help="0: run small QPs on the 512-thread instance too (A/B of the 256-thread instance); 2: on the 128-thread instance (seven workgroups per CU)")
    ap.add_argument("--narrow-rows", type=int, default=1, help="0: Schur assembly with one wavefront per column also for small QPs (A/B)")
    ap.add_argument("--place-panel-wave", type=int, default=-1, help="0: every workgroup runs its serial chains on wavefront 0; 1: panel waves placed on SIMDs 0 / 1; 2: + row ownership by SIMD (A/B; default: the library's)")
    ap.add_argument("--sweep-ranks", type=int, default=0, help="most ranks per sweep of the rank update: 16 or 32 (A/B; 0: library default = 16)")
    ap.add_argument("--kkt-compact", type=int, default=-1, help="KKT mode: 0 = factorise the whole (n+m) panel with its unit rows (A/B; default: the active rows only)")
    ap.add_argument("--ld-align", type=int, default=0, help="leading dimension of the factor panels rounded up to this many doubles (A/B; 0: library default = 16)")
    ap.add_argument("--max-slots", type=int, default=0, help="resident factor slots = concurrent workgroups (0: library default)")
    ap.add_argument("--coop", type=int, default=-2, help="coop mode (several workgroups per QP, host-chained kernels): 1 force, 0 never, -1 automatic (default: the library's setting)")
    ap.add_argument("--coop-max-batch", type=int, default=0, help="largest batch the automatic choice runs in coop mode (0: library default)")
    ap.add_argument("--sequential-rank-sums", type=int, default=-2, help="pivots of an update sweep as the running pivot d_r = d_{r-1} + p_r (1), as d_0 + prefix tree (0), or the library's choice (default)")
    ap.add_argument("--opt", action="append", default=[], metavar="NAME=VALUE", help="any integer context option of qpg_ctx_set_option (A/B runs)")
    ap.add_argument("--lib", default=None, help="A/B runs: path of another HIP build of the library (tools/evidence/gpu_ab.sh)")
    ap.add_argument("--traffic-json", default=None,
                    help="PMC summary written by tools/evidence/round_artifacts.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                         "command on the same box).  Without it roofline.traffic is null: a plain run measures no counters.  With it, "
                         "the figure is quoted only if the hash of the kernel sources AND of the built library recorded in the summary "
                         "equal this tree's and the workload matches")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------
# launcher: --gpus N without a torch.distributed environment
# ---------------------------------------------------------------------------------------------------------------
def launch_ranks(args, argv):
    """Start N fresh child processes (one per GPU) before anything here has initialised the GPU; relay rank 0's
    JSON line; non-zero exit if any rank fails.  Never re-execs a GPU-initialised process."""
    import socket
    n = args.gpus
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, stderr=None))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    line = None
    for ln in (out0 or b"").decode(errors="replace").splitlines():
        if ln.startswith("{"):
            line = ln
    if any(codes) or line is None:
        sys.stderr.write("bench.py: rank exit codes %s%s\n" % (codes, "" if line else " (no JSON line from rank 0)"))
        return 1 if not any(codes) else max(abs(c) for c in codes) or 1
    print(line)
    return 0


# ---------------------------------------------------------------------------------------------------------------
# byte model, per QP; fp64 = 8 B, int32 indices = 4 B
# ---------------------------------------------------------------------------------------------------------------
def byte_model(n, m, nnzA, nnzQ, nfac=None):
    nf = nfac or n                              # rows of the factor: n (Schur) or n + m (KKT panel)
    nnzL = nf * (nf + 1) // 2                   # dense-triangle storage of the factor (natural ordering, F5)
    b_solve = 2 * nnzL * 8 + 8 * n + 16 * n     # L twice + D + rhs/d
    b_spmv_A = nnzA * 12 + 4 * (n + 1) + 8 * (m + n)
    b_spmv_Q = nnzQ * 12 + 4 * (n + 1) + 16 * n
    b_vec = 8 * (22 * m + 18 * n) + 2 * (2 * m * 12)
    return dict(b_solve=b_solve, b_solve_forward=nnzL * 8, b_spmv_vec_newton=2 * b_spmv_A + b_spmv_Q + b_vec, b_spmv_vec_outer=b_spmv_A + b_vec,
                b_refactor=nnzL * 8 + (nnzQ + nnzA) * 12, b_sweep_entry=16)   # a touched entry of L is read and written once


def phase_bytes(model, st, iters):
    """algorithmic bytes of one QP's solve by phase, from the device-side work counters.  "solve" follows SURVEY.md section 8d (L streamed
    twice per solve); "solve_fused_away" is the part of it this kernel never moves: the forward substitutions that ride on the last
    update sweep (one pass over L each; the sweep's own bytes are counted by the sweep)"""
    n_newton = int(st.n_solve)
    return {
        "solve": n_newton * model["b_solve"],
        "solve_fused_away": int(getattr(st, "n_fused_solve", 0)) * model["b_solve_forward"],
        "spmv_vectors": n_newton * model["b_spmv_vec_newton"] + max(iters - n_newton, 0) * model["b_spmv_vec_outer"],
        "factor": (int(st.n_refactor) + int(st.n_factor_Q)) * model["b_refactor"],
        "update": int(st.sweep_entries) * model["b_sweep_entry"],   # sum over sweeps of nnz(L[:, J0:]), counted by the sweep itself
    }


def pmc_traffic(name, key, lib=None, explicit=None):
    """HBM bytes per launch from the committed PMC summary `profiles/r*/final/<name>` (tools/evidence/round_artifacts.sh: separate rocprofv3 --pmc
    passes, FETCH_SIZE x 2 + WRITE_SIZE per MI355X_MICROARCH.md), newest round first -- quoted ONLY when the hashes of the kernel sources and of
    the loaded library recorded in the summary equal this run's and the workload key (batch, n, m) matches; else (None, None)."""
    import glob
    cands = [explicit] if explicit else sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "final", name)), reverse=True)
    for cand in cands:
        try:
            with open(cand) as f:
                pj = json.load(f)
            if pj["source_sha256"] == source_sha256() and pj.get("lib_sha256") == lib_sha256(lib) and (pj["batch"], pj["n"], pj["m"]) == tuple(key):
                return float(pj["traffic_bytes_per_launch"]), os.path.relpath(cand, ROOT)
        except Exception:   # noqa: BLE001
            pass
    return None, None


def cpu_model_string():
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cholmod_probe():
    """the genuine CHOLMOD path is the CPU baseline only if the box has SuiteSparse."""
    import ctypes.util
    lib = ctypes.util.find_library("cholmod")
    hdr = [p for p in ("/usr/include/suitesparse/cholmod.h", "/usr/include/cholmod.h", "/usr/local/include/cholmod.h",
                       "/opt/conda/include/cholmod.h") if os.path.exists(p)]
    return {"libcholmod": lib, "cholmod_h": hdr[0] if hdr else None}


def write_problem_file(path, problems, settings_kw):
    """the sample of the batch in the flat binary layout oracle/cpu_bench.c reads (int64 / fp64, little endian)"""
    import ctypes as C
    import numpy as np
    from oracle import binding as ob
    st = ob.default_settings(**settings_kw)
    with open(path, "wb") as f:
        f.write(np.array([0x5150424e, len(problems)], dtype=np.int64).tobytes())
        f.write(bytes(memoryview(C.string_at(C.addressof(st), C.sizeof(st)))))
        for p in problems:
            f.write(np.array([p.n, p.m, int(p.Qp[-1]), int(p.Ap[-1])], dtype=np.int64).tobytes())
            for arr, dt in ((p.Qp, np.int64), (p.Qi, np.int64), (p.Qx, np.float64), (p.Ap, np.int64), (p.Ai, np.int64), (p.Ax, np.float64),
                            (p.q, np.float64), (np.array([getattr(p, "c", 0.0)]), np.float64), (p.bmin, np.float64), (p.bmax, np.float64)):
                f.write(np.ascontiguousarray(arr, dtype=dt).tobytes())


def cpu_baseline(problems, settings_kw, workload, budget_s=20.0, sparse=False):
    """Oracle ("port") on the host cores through a C pthread harness (oracle/cpu_bench.c: one workspace per thread, no
    Python in the timed loop); bounded sample.  Reported next to the GPU figure, not a target."""
    cores = max(1, min(os.cpu_count() or 1, 256))
    exe = os.path.join("/tmp", "qpalm_cpu_bench_%d" % os.getpid())
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "cpu_bench", "OUT=" + exe], stdout=subprocess.DEVNULL)
    pfile = exe + ".bin"
    try:
        def run(sample, threads, passes=1):
            write_problem_file(pfile, sample, settings_kw)
            r = subprocess.run([exe, pfile, str(threads), str(passes), "1" if sparse else "0"], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("cpu_bench failed (rc %d): %s %s" % (r.returncode, r.stdout[-300:], r.stderr[-300:]))
            return json.loads(r.stdout.strip().splitlines()[-1])
        probe = run(problems[:1], 1)                       # one QP alone on one core
        t_probe = probe["setup_plus_solve_s_per_qp"]
        # The thread count is calibrated: at n = 1000 every thread streams its own 8 MB factor, and with all 256 hardware
        # threads of the 2 x 64-core EPYC 9575F box busy a QP took 8.65 s instead of the 0.086 s it takes alone (30 QP/s
        # for the whole machine).  One short run per candidate count (a quarter, half, all), the best one is used.
        calib, best = {}, None
        for th in sorted({max(1, cores // 4), max(1, cores // 2), cores}):
            r = run(problems[:min(len(problems), th)], th, 1)
            calib[th] = round(r["qps"], 3)
            if best is None or r["qps"] > best[1]["qps"]:
                best = (th, r)
        threads, t_loaded = best[0], best[1]["setup_plus_solve_s_per_qp"]
        # ~budget_s of wall time with those threads busy
        nsample = int(min(len(problems), threads * max(1, int(budget_s / max(t_loaded, 1e-6)))))
        passes = int(max(1, min(64, budget_s / max(nsample * t_loaded / threads, 1e-6))))   # small QPs: the sample several times
        res = run(problems[:nsample], threads, passes)
        cores = threads
    finally:
        for fpath in (exe, pfile):
            try:
                os.remove(fpath)
            except OSError:
                pass
    probe_lib = cholmod_probe()
    have = bool(probe_lib["libcholmod"] and probe_lib["cholmod_h"])
    return {"value": res["qps"], "unit": "QP/s", "cores": cores, "kind": "port", "cpu_model": cpu_model_string(),
            "setup_plus_solve_s_per_qp": res["setup_plus_solve_s_per_qp"], "solve_s_per_qp": res["solve_s_per_qp"],
            "single_qp_alone_s": t_probe, "iter_mean": res["iter_mean"], "wall_s": res["wall_s"], "threads_tried_qps": calib,
            "cholmod_probe": probe_lib,
            "sample": "%d of the batch's %s QPs x %d pass(es), setup + solve timed as info.run_time does (eps 1e-6), %d pthreads (the fastest of the thread counts in threads_tried_qps) pulling QPs from a "
                      "shared counter (oracle/cpu_bench.c, no Python in the loop), oracle/qpalm_oracle.c (%s) "
                      "built -O3 -march=native; one QP alone on one core: %.4f s; %s"
                      % (nsample, workload, passes, cores, "sparse-storage L D L', natural ordering, path updates" if sparse else "dense LDL', rank-1 sweeps applied 8 at a time per pass over L (bit-identical to one at a time)", t_probe,
                         "a system CHOLMOD is present (probe in cholmod_probe) but this figure is the restatement" if have else
                         "no system CHOLMOD on the box (probe in cholmod_probe), so this is the restatement, not CHOLMOD")}


def kkt_spot_check(probs, xs, ys, idx, bmin_all=None, bmax_all=None):
    """numpy KKT residuals of a few QPs of the timed batch on the unscaled data (a broken build must not post a number)"""
    import numpy as np
    import scipy.sparse as sp
    worst = 0.0
    for k in idx:
        p = probs[k]
        A = sp.csc_matrix((p.Ax, p.Ai, p.Ap), shape=(p.m, p.n))
        Ql = sp.csc_matrix((p.Qx, p.Qi, p.Qp), shape=(p.n, p.n))
        Q = Ql + sp.tril(Ql, -1).T
        x, y = xs[k], ys[k]
        ax = A @ x
        bmin = p.bmin if bmin_all is None else bmin_all[k]   # the bounds of the LAST step (update_bounds moved them)
        bmax = p.bmax if bmax_all is None else bmax_all[k]
        prim = np.max(np.maximum(bmin - ax, 0) + np.maximum(ax - bmax, 0)) / max(1.0, np.max(np.abs(ax)))
        Qx, Aty = Q @ x, A.T @ y
        dual = np.max(np.abs(Qx + p.q + Aty)) / max(1.0, np.max(np.abs(Qx)), np.max(np.abs(p.q)), np.max(np.abs(Aty)))
        worst = max(worst, prim, dual)
    return worst


def mpc_problems(B, rank, rng):
    """mpc-160: one plant per 64 QPs, every QP its own initial state; the steps move the initial state (bounds of the x_0 rows)"""
    from qpalm_amd.problems import random_mpc_qp
    nx, nu, T = 10, 5, 10
    plants, probs = {}, []
    for k in range(B):
        seed = rank * 100003 + k // 64
        if seed not in plants:
            plants[seed] = random_mpc_qp(T=T, nx=nx, nu=nu, seed=seed)
        base = plants[seed]
        x0 = 2.0 * (2 * rng.random(nx) - 1)
        bmin, bmax = base.bmin.copy(), base.bmax.copy()
        bmin[:nx] = x0
        bmax[:nx] = x0
        probs.append(type(base)(base.n, base.m, base.Qp, base.Qi, base.Qx, base.Ap, base.Ai, base.Ax, base.q, bmin, bmax))
    return probs


def mpc160_line(ctx, B, kkt, steps, warmup=1, traffic_dir=None):
    """BASELINE.json config 3 next to the headline: B mpc-160 QPs through a warm-started receding-horizon
    sequence (update_bounds + warm start from the previous solution, simulations/randomMPCsequential.m:158-177), Schur panel with
    rank updates or (kkt) the (n+m) x (n+m) KKT panel with row additions / deletions.  Returns the sub-object of the JSON line."""
    import numpy as np
    from qpalm_amd.solver import QpalmBatch
    rng = np.random.default_rng(777)
    probs = mpc_problems(B, 0, rng)
    n, m = probs[0].n, probs[0].m
    kw = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0)
    if kkt:
        kw["factorization_method"] = 0
    bt = QpalmBatch(ctx, probs, ctx.default_settings(**kw))
    bmin_all, bmax_all = ctx.pinned_array((B, m)), ctx.pinned_array((B, m))
    bmin_all[:] = np.stack([p.bmin for p in probs])
    bmax_all[:] = np.stack([p.bmax for p in probs])
    sol = (ctx.pinned_array((B, n)), ctx.pinned_array((B, m)))
    first = [True]

    def step():
        if first[0]:
            bt.warm_start(None, None)
            first[0] = False
        else:
            x0 = bmin_all[:, :10] + 0.1 * rng.standard_normal((B, 10))
            bmin_all[:, :10] = x0
            bmax_all[:, :10] = x0
            if bt.update_bounds(bmin_all, bmax_all) != 0:
                raise RuntimeError("update_bounds rejected the new bounds")
            bt.warm_start_last()
        bt.solve()
        bt.solution(out=sol)
    for _ in range(warmup):
        step()
    kms = []
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
        kms.append(bt.last_solve_ms())
    dt = time.perf_counter() - t0
    infos, stats = bt.infos(), bt.stats_all()
    st = np.array([int(i.status_val) for i in infos])
    it = np.array([int(i.iter) for i in infos])
    worst = kkt_spot_check(probs, sol[0], sol[1], sorted({0, B // 2, B - 1}), bmin_all, bmax_all)
    mean = lambda f: float(np.mean([f(s) for s in stats]))
    # byte model of config 3: the same per-unit bytes as the headline at this size -- per QP a
    # factorisation writes nnz(L) 8 = 103 KB (Schur: 160 rows; KKT: the (n+m)-row panel) and reads Q, A once, a Newton solve streams L twice
    # (minus the forward halves fused into a sweep), a sweep reads and writes the entries it touches (device counter), SpMVs and vectors as 8d
    # -- counted from the device-side work counters of the LAST warm-started step, over that step's kernel time (HIP events on the launch's stream)
    tot = {"solve": 0, "spmv_vectors": 0, "factor": 0, "update": 0, "solve_fused_away": 0}
    for b in range(B):
        pb = phase_bytes(byte_model(n, m, int(probs[b].Ap[-1]), int(probs[b].Qp[-1]), (n + m) if kkt else None), stats[b], int(it[b]))
        for k in tot:
            tot[k] += pb[k]
    fused_away = tot.pop("solve_fused_away")
    alg = sum(tot.values()) - fused_away
    tname = "mpc160_kkt_pmc_traffic.json" if kkt else "mpc160_pmc_traffic.json"
    traffic, traffic_src = pmc_traffic(tname, (B, n, m), None, os.path.join(traffic_dir, tname) if traffic_dir else None)
    roof = {"bound": "hbm", "kernel": "k_solve (persistent, one workgroup per QP), one warm-started step", "achieved": alg / (kms[-1] * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": alg / (kms[-1] * 1e-3) / 1e9 / HBM_PEAK_GBS, "kernel_ms": kms[-1], "algorithmic_bytes_per_launch": alg, "fused_away_bytes_per_launch": fused_away,
            "bytes_per_qp": {k: v / B for k, v in tot.items()}, "traffic": traffic, "traffic_source": traffic_src,
            "traffic_over_algorithmic": (traffic / alg) if traffic else None,
            "note": "latency-bound at this size: every phase of a 160-variable QP is a chain of dependent round trips; the fraction says how far "
                    "the bytes it must move are from what the chip could stream, the traffic ratio how many bytes it moves beyond them"}
    out = {"workload": "mpc-160 (n=%d m=%d), %d QPs, %d warm-started steps, %s" % (n, m, B, steps, "KKT panel, row add / delete" if kkt else "Schur panel, rank updates"),
           "value": B * steps / dt, "unit": "QP/s", "ms_per_step": 1e3 * dt / steps, "kernel_ms_per_step": float(np.mean(kms)), "roofline": roof,
           "all_solved": bool(np.all(st == 1)), "kkt_spot_check_worst_rel": worst, "iter_mean": float(it.mean()),
           "per_qp_mean": {k: mean(lambda s, k=k: getattr(s, k)) for k in ("n_refactor", "n_factor_Q", "n_sweeps", "n_rank1", "n_solve")},
           "ms_per_qp_in_kernel": {"total": mean(lambda s: s.ms_total), "factor": mean(lambda s: s.ms_factor), "update": mean(lambda s: s.ms_update),
                                   "solve": mean(lambda s: s.ms_solve), "linesearch": mean(lambda s: s.ms_linesearch)}}
    bt.close()
    return out


def sparse_workload(args, ctx, rank, world, dist, torch):
    """`--workload sparse-banded-2000 | sparse-blocks-2000`: B (default 2048) sparse QPs with n = 2000 per GPU on the sparse L D L'
    (qpalm_sparse.h: one workgroup per QP, nested-dissection ordering where the natural elimination tree is a chain).  The kernel is
    bound by chains of dependent round trips, not by bytes: the line carries the phase split instead of a roofline."""
    import numpy as np
    from qpalm_amd.problems import sparse_qp
    from qpalm_amd.solver import QpalmBatch
    kind = args.workload.split("-")[1]
    n = args.n or 2000
    B = args.batch if args.batch != 8192 else 2048
    ctx.set_option("sparse_factor", 1)
    settings_kw = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0)
    distinct = [sparse_qp(n, kind, seed=700 + 64 * rank + k) for k in range(64)]
    probs = [distinct[k % 64] for k in range(B)]
    bt = QpalmBatch(ctx, probs, ctx.default_settings(**settings_kw))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        bt.warm_start(None, None)
        bt.solve()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda:%d" % int(os.environ.get("LOCAL_RANK", "0")), dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    infos, stats = bt.infos(), bt.stats_all()
    xs, ys = bt.solution()
    ok = all(int(i.status_val) == 1 for i in infos) and kkt_spot_check(probs, xs, ys, sorted({0, B // 3, B // 2, B - 1})) <= 1e-4
    mean = lambda f: float(np.mean([f(s) for s in stats]))
    perm, levels = bt.sparse_perm(0)
    nnzL, nbytes = bt.sparse_info(0)
    if rank != 0:
        return 0 if ok else 1
    out = {"metric": "QP/s", "value": world * B * args.steps / elapsed, "unit": "QP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "%s: batch of %d sparse QPs per GPU (%s pattern, n=%d m=%d), sparse L D L' with %d levels (ordering: %s), nnz(L)=%d, eps 1e-6, cold start"
                                  % (args.workload, B, kind, probs[0].n, probs[0].m, levels, "natural" if np.array_equal(perm, np.arange(len(perm))) else "nested dissection", nnzL),
                      "batch_per_gpu": B, "n": probs[0].n, "m": probs[0].m, "kernel": "k_solve<sparse>", "parallelism": "batch-shard x%d" % world, "device_block_MB": nbytes / 2 ** 20},
           "roofline": None,
           "solve_stats": {"all_solved": bool(ok), "iter_mean": float(np.mean([int(i.iter) for i in infos])),
                           "per_qp_mean": {k: mean(lambda s, k=k: getattr(s, k)) for k in ("n_refactor", "n_factor_Q", "n_rank1", "n_solve")},
                           "ms_per_qp_in_kernel": {"total": mean(lambda s: s.ms_total), "factor": mean(lambda s: s.ms_factor), "update": mean(lambda s: s.ms_update), "solve": mean(lambda s: s.ms_solve),
                                                   "linesearch": mean(lambda s: s.ms_linesearch), "residuals": mean(lambda s: s.ms_dbg[12])},
                           "note": "latency-bound (chains of dependent round trips per column and per level), no byte model: roofline is null for this workload"}}
    if not args.no_cpu:
        try:
            out["cpu_baseline"] = cpu_baseline(distinct, settings_kw, args.workload, budget_s=15.0, sparse=True)
        except Exception as e:   # noqa: BLE001
            out["cpu_baseline"] = {"error": repr(e)[:300]}
    bt.close()
    print(json.dumps(out))
    return 0 if ok else 1


# ---------------------------------------------------------------------------------------------------------------
def worker(args):
    import numpy as np
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d\n" % (args.gpus, world))
        return 2
    if not torch.cuda.is_available() or local >= torch.cuda.device_count():
        sys.stderr.write("bench.py: rank %d: no HIP device %d visible (this benchmark has no CPU path)\n" % (rank, local))
        return 3
    dist = None
    torch.cuda.set_device(local)
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from qpalm_amd.problems import random_mpc_qp, random_qp
    from qpalm_amd.solver import Context, QpalmBatch
    ctx = Context(local, lib_path=args.lib)
    ctx.set_option("update_rank_threshold", args.rank_threshold)
    if args.max_slots:
        ctx.set_option("max_slots", args.max_slots)
    if args.ld_align:
        ctx.set_option("ld_align", args.ld_align)
    if args.sweep_ranks:
        ctx.set_option("sweep_ranks", args.sweep_ranks)
    if args.coop > -2:
        ctx.set_option("coop", args.coop)
    if args.coop_max_batch:
        ctx.set_option("coop_max_batch", args.coop_max_batch)
    if args.kkt_compact >= 0:
        ctx.set_option("kkt_compact", args.kkt_compact)
    if args.small_workgroups != 1:
        ctx.set_option("small_workgroups", args.small_workgroups)
    if args.place_panel_wave >= 0:
        ctx.set_option("place_panel_wave", args.place_panel_wave)
    if not args.narrow_rows:
        ctx.set_option("narrow_rows", 0)
    if args.sequential_rank_sums > -2:
        ctx.set_option("sequential_rank_sums", args.sequential_rank_sums)
    for kv in args.opt:
        name, _, val = kv.partition("=")
        ctx.set_option(name, int(val))
    B = args.batch
    if args.workload.startswith("sparse-"):
        return sparse_workload(args, ctx, rank, world, dist, torch)
    settings_kw = dict(eps_abs=1e-6, eps_rel=1e-6, verbose=0)
    if args.kkt:
        settings_kw["factorization_method"] = 0
    rng = np.random.default_rng(12345 + rank)
    if args.workload == "random-1000":
        n = args.n or 1000
        m = args.m or 2 * n
        dens_A = 0.01 if n >= 400 else max(0.01, 4.0 / n)
        dens_M = 0.005 if n >= 400 else max(0.005, 2.0 / n)
        probs = [random_qp(n, m, seed=1000 + rank * B + k, density_A=dens_A, density_M=dens_M) for k in range(B)]
        wl = "random-%d: batch of %d QPs per GPU, n=%d m=%d nnz(A)~%d nnz(tril Q)~%d, eps 1e-6, scaling 10, cold start" % (
            n, B, n, m, int(probs[0].Ap[-1]), int(probs[0].Qp[-1]))
    else:
        probs = mpc_problems(B, rank, rng)
        n, m = probs[0].n, probs[0].m
        wl = "mpc-160: batch of %d MPC QPs per GPU (T=10, nx=10, nu=5: n=%d m=%d nnz(A)=%d), eps 1e-6, scaling 10; every step moves " \
             "the initial states (update_bounds) and warm-starts from the previous solution" % (B, n, m, int(probs[0].Ap[-1]))
    bt = QpalmBatch(ctx, probs, ctx.default_settings(**settings_kw))   # upload + Ruiz scaling: not timed
    # the arrays that cross PCIe every step of the MPC loop live in page-locked host memory (qpg_host_alloc)
    bmin_all, bmax_all = ctx.pinned_array((B, m)), ctx.pinned_array((B, m))
    bmin_all[:] = np.stack([p.bmin for p in probs])
    bmax_all[:] = np.stack([p.bmax for p in probs])
    sol_out = (ctx.pinned_array((B, n)), ctx.pinned_array((B, m))) if args.workload == "mpc-160" else None
    state = {"x": None, "y": None}

    def gather_results():
        if dist is None:
            return
        from qpalm_amd.dist import device_view, info_matrix
        dev = "cuda:%d" % local
        tx = device_view(bt, "solution_x", (B, n), dev)   # zero-copy views of the HBM arrays
        ty = device_view(bt, "solution_y", (B, m), dev)
        ti = torch.from_numpy(info_matrix(bt)).to(dev)     # QPALMInfo records (include/types.h:76-95), one D2H copy
        for t in (tx, ty, ti):
            g = [torch.empty_like(t) for _ in range(world)] if rank == 0 else None
            dist.gather(t, g, dst=0)

    def step():
        if args.workload == "mpc-160":
            if state["x"] is not None:   # the plant moved: new initial state, warm start from the previous solution
                x0 = bmin_all[:, :10] + 0.1 * rng.standard_normal((B, 10))
                bmin_all[:, :10] = x0
                bmax_all[:, :10] = x0
                if bt.update_bounds(bmin_all, bmax_all) != 0:
                    raise RuntimeError("update_bounds rejected the new bounds: " + ctx.L.qpg_last_error().decode())
                bt.warm_start_last()      # previous solution, straight from HBM (== warm_start(*solution()), tests/test_mpc_scale.py)
            else:
                bt.warm_start(None, None)
        else:
            bt.warm_start(None, None)
        bt.solve()
        if args.workload == "mpc-160":
            state["x"], state["y"] = bt.solution(out=sol_out)
        gather_results()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        step()
        kernel_ms.append(bt.last_solve_ms())
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda:%d" % local, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- checks + accounting of the last step (outside the timed region) ------------------------------------
    infos, stats = bt.infos(), bt.stats_all()
    statuses = np.array([int(i.status_val) for i in infos])
    iters = np.array([int(i.iter) for i in infos])
    n_bad = int(np.sum(statuses != 1))
    xs, ys = bt.solution()
    import hashlib
    sol_hash = hashlib.sha256(np.ascontiguousarray(xs).tobytes() + np.ascontiguousarray(ys).tobytes()).hexdigest()[:16]   # same-box A/B: equal results?
    kkt = kkt_spot_check(probs, xs, ys, sorted({0, B // 3, B // 2, B - 1}), bmin_all, bmax_all)
    ok = (n_bad == 0) and (kkt <= 1e-4)
    tot = {"solve": 0, "spmv_vectors": 0, "factor": 0, "update": 0, "solve_fused_away": 0}
    for b in range(B):
        model = byte_model(n, m, int(probs[b].Ap[-1]), int(probs[b].Qp[-1]), (n + m) if args.kkt else None)
        pb = phase_bytes(model, stats[b], int(iters[b]))
        for k in tot:
            tot[k] += pb[k]
    fused_away = tot.pop("solve_fused_away")
    model_bytes = sum(tot.values())           # SURVEY.md section 8d, unfused algorithm
    tot_bytes = model_bytes - fused_away      # what this kernel has to move: the figure the roofline fraction is quoted on
    kms = float(np.mean(kernel_ms))
    achieved = tot_bytes / (kms * 1e-3) / 1e9
    mean = lambda f: float(np.mean([f(s) for s in stats]))
    phase_ms = {"total": mean(lambda s: s.ms_total), "factor": mean(lambda s: s.ms_factor), "update": mean(lambda s: s.ms_update),
                "solve": mean(lambda s: s.ms_solve), "linesearch": mean(lambda s: s.ms_linesearch), "residuals": mean(lambda s: s.ms_dbg[12])}
    phase_ms["dbg"] = [mean(lambda s, k=k: s.ms_dbg[k]) for k in range(16)]
    conc, wg_threads, wg_lds = bt.launch_shape()
    # NORMALISED GB/s of a phase = bytes of all QPs / (time the phase occupies one of `conc` concurrent workgroups): what the chip would move if
    # every resident workgroup were in that phase at once -- a normalisation (it can exceed what the memory system delivers), not a measurement
    def phase_gbs(nbytes, ms_per_qp):
        return nbytes / (max(ms_per_qp, 1e-9) * 1e-3 * B / conc) / 1e9
    phases = {
        "solve": {"bytes": tot["solve"] - fused_away, "ms_per_qp": phase_ms["solve"], "GBps_normalised": phase_gbs(tot["solve"] - fused_away, phase_ms["solve"]),
                  "fused_away_bytes": fused_away},
        "update": {"bytes": tot["update"], "ms_per_qp": phase_ms["update"], "GBps_normalised": phase_gbs(tot["update"], phase_ms["update"])},
        "factor": {"bytes": tot["factor"], "ms_per_qp": phase_ms["factor"], "GBps_normalised": phase_gbs(tot["factor"], phase_ms["factor"]),
                   "flop": float(sum((int(s.n_refactor) + int(s.n_factor_Q)) for s in stats)) * n ** 3 / 3.0,
                   "reread_bytes": float(sum(int(s.factor_reread_entries) for s in stats)) * 8},
        "spmv_vectors": {"bytes": tot["spmv_vectors"], "ms_per_qp": phase_ms["linesearch"] + phase_ms["residuals"],
                         "GBps_normalised": phase_gbs(tot["spmv_vectors"], phase_ms["linesearch"] + phase_ms["residuals"])},
    }
    phases["factor"]["TFLOPs"] = phases["factor"]["flop"] / (max(phase_ms["factor"], 1e-9) * 1e-3 * B / conc) / 1e12
    rc = 0
    if rank == 0:
        copy_gbs = ctx.hbm_copy_gbs(1 << 30, 5)             # attainable ceilings on this box: copy (read + write) ...
        read_gbs = ctx.hbm_read_gbs(1 << 30, 5)             # ... and a read-only stream
        model0 = byte_model(n, m, int(probs[0].Ap[-1]), int(probs[0].Qp[-1]), (n + m) if args.kkt else None)
        nsl = conc
        ms_ldl = bt.ldlsolve_all(reps=4)                    # stand-alone LDL' solve kernel ("HBM GB/s on LDL")
        ldl_bytes = nsl * model0["b_solve"]
        # the PMC summary of the SAME build (tools/evidence/round_artifacts.sh writes it, separate rocprofv3 --pmc passes): --traffic-json names
        # one; without the flag the committed summaries under profiles/ are tried, newest round first.  Quoted only when the hashes of the
        # kernel sources and of the loaded library recorded in the summary equal this run's (else roofline.traffic stays null).
        traffic, traffic_src = pmc_traffic("k_solve_pmc_traffic.json", (B, n, m), args.lib, args.traffic_json) if world == 1 else (None, None)
        out = {
            "metric": "QP solves/sec (batched %s); `value` = solve only, batch resident in HBM -- with qpalm_setup in the clock as the reference's run_time has it: `value_setup_plus_solve`"
                      % ("random n=%d,m=%d" % (n, m) if args.workload == "random-1000" else "MPC n=%d,m=%d, warm-started sequence" % (n, m)),
            "value": world * B * args.steps / elapsed, "unit": "QP/s",
            # the reference's info.run_time = setup_time + solve_time (src/qpalm.c:493-495,721-723): the same rate with qpalm_setup's work
            # of this batch added to one step (set_problem: host-side copies / format conversion, single-threaded; batch_setup: packing,
            # upload over PCIe, Ruiz scaling + derived copies on the device); `value` is the solve rate with the batch resident in HBM
            "value_setup_plus_solve": world * B / (elapsed / args.steps + bt.set_problem_s + bt.batch_setup_s),
            "setup": {"set_problem_s": bt.set_problem_s, "batch_setup_s": bt.batch_setup_s,
                      "value_device_setup_plus_solve": world * B / (elapsed / args.steps + bt.batch_setup_s)},
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "gathered_over": ("rccl, %d rank(s)" % world) if dist is not None else None,
            "config": {"workload": wl, "batch_per_gpu": B, "n": n, "m": m, "kernel": "k_solve<%d>" % (lambda r: 1 if r <= 1 else (2 if r <= 2 else (4 if r <= 4 else 0)))(-(-(n + (m if args.kkt else 0)) // wg_threads)),
                       "factorization": "kkt" if args.kkt else "schur", "parallelism": "batch-shard x%d" % world,
                       "update_rank_threshold": args.rank_threshold,
                       "workgroups": conc, "threads_per_workgroup": wg_threads, "lds_per_workgroup": wg_lds},
            "roofline": {"bound": "hbm", "kernel": "k_solve (persistent, one workgroup per QP)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_src, "kernel_ms": kms, "algorithmic_bytes_per_launch": tot_bytes,
                         # SURVEY.md section 8d counts L twice per Newton solve; the forward halves that ride on the last update sweep are
                         # never streamed by this kernel, so they are NOT in `achieved` / `frac`; the unfused model is quoted beside it
                         "fused_away_bytes_per_launch": fused_away, "survey_model_bytes_per_launch": model_bytes,
                         "frac_survey_model": model_bytes / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "traffic_over_algorithmic": (traffic / tot_bytes) if traffic else None,
                         "guide_copy_GBps": 6290.0,   # MI355X_MICROARCH.md: float4 copy, 79 % of the 8 TB/s spec figure
                         "measured_copy_GBps": copy_gbs, "measured_read_GBps": read_gbs, "frac_of_measured_copy": achieved / copy_gbs,
                         # the HBM bytes the launch really moved (PMC) per second, against the same yardsticks: how far the
                         # kernel is from the bandwidth this chip delivers to a plain copy
                         "traffic_GBps": (traffic / (kms * 1e-3) / 1e9) if traffic else None,
                         "traffic_frac_of_measured_copy": (traffic / (kms * 1e-3) / 1e9 / copy_gbs) if traffic else None,
                         "bytes_note": "update bytes = 16 B x entries of L[:, J0:] actually swept (device counter), not the 8d upper bound; solve bytes = "
                                       "8d's two passes over L minus the forward passes fused into a sweep (device counter n_fused_solve)",
                         "phases": phases},
            "ldl_solve": {"kernel": "k_ldlsolve_all", "qps": nsl, "ms": ms_ldl, "bytes": ldl_bytes,
                          "achieved": ldl_bytes / (ms_ldl * 1e-3) / 1e9, "unit": "GB/s",
                          "frac": ldl_bytes / (ms_ldl * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "frac_of_measured_read": ldl_bytes / (ms_ldl * 1e-3) / 1e9 / read_gbs},
            "solve_stats": {"all_solved": n_bad == 0, "kkt_spot_check_worst_rel": kkt, "solution_sha256_16": sol_hash,
                            "iter_mean": float(iters.mean()), "iter_max": int(iters.max()),
                            "per_qp_mean": {k: mean(lambda s, k=k: getattr(s, k)) for k in ("n_refactor", "n_factor_Q", "n_sweeps", "n_rank1", "n_solve", "sweep_entries")},
                            # the update sweeps' per-column guard (qp_rank_pivots): columns summed again as the reference's running pivot because a pivot
                            # shrank by 2^8 or more inside a sweep, out of all columns of the diagonal-block recurrences; Newton steps redone with a
                            # fresh factorisation because the direction out of an updated factor was not finite
                            "pivot_guard": {"columns_resummed": int(sum(int(s.n_seq_columns) for s in stats)), "columns": int(sum(int(s.n_sweep_columns) for s in stats)),
                                            "newton_steps_redone": int(sum(int(s.n_guard_refactor) for s in stats))},
                            "phase_ms_per_qp": phase_ms},
        }
        if world == 1 and args.workload == "random-1000" and not args.no_mpc and not args.kkt and not args.n and not args.lib:
            # config 3 in the driver's record: Schur and KKT mode, a few warm-started steps each (outside the timed region of the headline)
            try:
                tdir = os.path.dirname(os.path.abspath(args.traffic_json)) if args.traffic_json else None   # (the evidence run: the summaries sit next to the headline's)
                out["mpc160"] = {"schur": mpc160_line(ctx, 8192, False, 5, traffic_dir=tdir), "kkt": mpc160_line(ctx, 2048, True, 3, traffic_dir=tdir)}
            except Exception as e:   # never lose the headline line over the side figures
                out["mpc160"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(probs, settings_kw, args.workload)
        else:
            out["cpu_baseline"] = None
        if not ok:
            sys.stderr.write("bench.py: INVALID RUN: %d QPs not solved, KKT spot check %.3e\n" % (n_bad, kkt))
            out["invalid"] = "%d QPs not solved, KKT spot check %.3e" % (n_bad, kkt)
        print(json.dumps(out))
    if not ok:
        rc = 4
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return rc


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args, argv)      # before torch / HIP are touched in this process
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
