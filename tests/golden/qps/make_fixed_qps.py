#!/usr/bin/env python3
"""Generator of the synthetic FIXED-format QPS fixtures under tests/golden/qps/fixed/ (data, not reference text).

The Maros-Meszaros set is distributed in the old fixed-column MPS/QPS format (names may contain blanks); it is not in the
reference tree and there is no network, so these 50 files stand in for it: convex QPs of Maros-Meszaros-like shapes (n = 3..48,
box / one-sided / equality rows, RANGES incl. negative ranges on E rows, FR / MI / PL / UP / LO / FX bounds, an objective constant
through a negative RHS on the objective row, one or two entries per COLUMNS line, comment lines, names with embedded blanks).

`problem(k)` returns the exact data file k encodes (every number is a small multiple of 1/8 and printed exactly), so a test can
compare the C reader's output with it entry by entry.  Run as a script to (re)write the files.
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "fixed")
COUNT = 50
INF = 1e20


def _num(v):
    s = ("%.6f" % v).rstrip("0").rstrip(".")
    if s in ("-0", ""):
        s = "0"
    if "." not in s:
        s += "."
    assert float(s) == v and len(s) <= 12, (v, s)
    return s


def _name(prefix, k, blank):
    """8-character names; with `blank` they contain embedded blanks like 'C      7' (legal only in the fixed format)"""
    if blank:
        return "%s%7d" % (prefix, k)
    return "%s%d" % (prefix, k)


def problem(k):
    """file k: a convex QP that is feasible by construction (a point x* inside the variable bounds, row bounds placed around A x*)
    except for every tenth file, whose first equality row is moved away from A x* by 1000 (infeasibility detection)"""
    rng = np.random.default_rng(7000 + k)
    n = int(rng.integers(3, 49))
    mr = int(rng.integers(1, max(2, (3 * n) // 2)))
    blank = (k % 5) != 4                       # every fifth file has plain names but still fixed columns
    dens = min(1.0, 3.0 / n + 0.05)
    A = np.zeros((mr, n))
    for i in range(mr):
        cols = rng.choice(n, size=max(1, int(rng.binomial(n, dens))), replace=False)
        A[i, cols] = rng.integers(-8, 9, size=len(cols)) / 4.0
    M = np.zeros((n, n))
    for j in range(n):
        for i in range(j + 1, n):
            if rng.random() < min(1.0, 2.0 / n):
                M[i, j] = rng.integers(-8, 9) / 8.0
    Q = M + M.T
    Q += np.diag(np.sum(np.abs(Q), axis=1) + rng.integers(1, 9, size=n) / 8.0)    # strictly diagonally dominant: convex
    q = rng.integers(-24, 25, size=n) / 8.0
    btype, xs = {}, np.zeros(n)
    for j in range(n):
        u = rng.random()
        lo, hi = 0.0, 4.0                      # default bound [0, 1e20]
        if u < 0.15:
            btype[j] = ("FR", None); lo = -4.0
        elif u < 0.25:
            btype[j] = ("MI", None); lo = -4.0
        elif u < 0.45:
            v = rng.integers(1, 33) / 8.0
            btype[j] = ("UP", v); hi = v
        elif u < 0.55:
            v = rng.integers(-16, 1) / 8.0
            btype[j] = ("LO", v); lo = v
        elif u < 0.60:
            v = rng.integers(-8, 9) / 8.0
            btype[j] = ("FX", v); lo = hi = v
        elif u < 0.65:
            btype[j] = ("PL", None)
        xs[j] = lo if lo == hi else np.clip(np.round(rng.uniform(lo, hi) * 4) / 4, np.ceil(lo * 4) / 4, np.floor(hi * 4) / 4)
    ax = A @ xs                                # multiples of 1/16
    types = rng.choice(list("LGE"), size=mr, p=[0.4, 0.3, 0.3])
    rhs, ranges = np.zeros(mr), {}
    for i in range(mr):
        slack = rng.integers(0, 9) / 4.0
        rng_i = rng.integers(1, 17) / 8.0 if rng.random() < 0.25 else None
        t = types[i]
        if rng_i is not None:
            slack = min(slack, rng_i)
            if t == "E" and rng.random() < 0.5:
                rng_i = -rng_i
            ranges[i] = rng_i
        if t == "L":
            rhs[i] = ax[i] + slack
        elif t == "G":
            rhs[i] = ax[i] - slack
        elif rng_i is None:
            rhs[i] = ax[i]
        elif rng_i >= 0:
            rhs[i] = ax[i] - slack             # [rhs, rhs + r]
        else:
            rhs[i] = ax[i] + slack             # [rhs + r, rhs]
    if k % 10 == 9:                            # an infeasible member: two contradicting equality rows on the same variables
        A = np.vstack([A, A[:1]])
        types = np.append(types, "E")
        e0 = np.where(types[:mr] == "E")[0]
        rhs = np.append(rhs, ax[0] + 1000.0)
        types[0] = "E"; rhs[0] = ax[0]; ranges.pop(0, None)
        mr += 1
    c0 = float(rng.integers(-8, 9) / 8.0) if k % 3 == 0 else 0.0
    return dict(k=k, n=n, mr=mr, blank=blank, A=A, Q=Q, q=q, types=types, rhs=rhs, ranges=ranges, c0=c0, btype=btype)


def expected(P):
    """the QP the reader must produce (A with the bound rows appended; MPS semantics as documented in include/qpalm_qps.h)"""
    n, mr = P["n"], P["mr"]
    free = [j for j in range(n) if P["btype"].get(j, ("", None))[0] == "FR"]
    brow, nxt = {}, mr
    for j in range(n):
        if j not in free:
            brow[j] = nxt
            nxt += 1
    m = nxt
    A = np.zeros((m, n))
    A[:mr] = P["A"]
    bmin, bmax = np.zeros(m), np.zeros(m)
    for i in range(mr):
        t, v = P["types"][i], P["rhs"][i]
        if t == "L":
            bmin[i], bmax[i] = -INF, v
        elif t == "G":
            bmin[i], bmax[i] = v, INF
        else:
            bmin[i], bmax[i] = v, v
        if i in P["ranges"]:
            r = P["ranges"][i]
            if t == "L":
                bmin[i] = bmax[i] - r
            elif t == "G":
                bmax[i] = bmin[i] + r
            elif r >= 0:
                bmax[i] = bmin[i] + r
            else:
                bmin[i] = bmax[i] + r
    for j, r in brow.items():
        A[r, j] = 1.0
        bmin[r], bmax[r] = 0.0, INF
        t, v = P["btype"].get(j, ("", None))
        if t == "UP":
            bmax[r] = v
        elif t == "LO":
            bmin[r] = v
        elif t == "FX":
            bmin[r] = bmax[r] = v
        elif t == "MI":
            bmin[r] = -INF
    return dict(n=n, m=m, A=A, Q=P["Q"], q=P["q"], bmin=bmin, bmax=bmax, c=P["c0"])


def _line(f1="", f2="", f3="", f4="", f5="", f6=""):
    s = " %-2s %-8s  %-8s  %12s" % (f1, f2, f3, f4)
    if f5:
        s += "   %-8s  %12s" % (f5, f6)
    return s.rstrip() + "\n"


def write(P, path):
    n, mr, blank = P["n"], P["mr"], P["blank"]
    rn = lambda i: _name("R", i + 1, blank)
    cn = lambda j: _name("C", j + 1, blank)
    obj = "OBJ" if not blank else "OBJ    F"
    with open(path, "w") as f:
        f.write("NAME          FIX%04d\n" % P["k"])
        f.write("* synthetic fixed-format fixture %d (tests/golden/qps/make_fixed_qps.py)\n" % P["k"])
        f.write("ROWS\n")
        f.write(_line("N", obj))
        for i in range(mr):
            f.write(_line(P["types"][i], rn(i)))
        f.write("COLUMNS\n")
        for j in range(n):
            ent = []
            if P["q"][j] != 0:
                ent.append((obj, P["q"][j]))
            ent += [(rn(i), P["A"][i, j]) for i in range(mr) if P["A"][i, j] != 0]
            if not ent:
                ent = [(obj, 0.0)]
            k = 0
            while k < len(ent):
                if k + 1 < len(ent) and (j + k) % 2 == 0:      # two entries on a line, sometimes
                    f.write(_line("", cn(j), ent[k][0], _num(ent[k][1]), ent[k + 1][0], _num(ent[k + 1][1])))
                    k += 2
                else:
                    f.write(_line("", cn(j), ent[k][0], _num(ent[k][1])))
                    k += 1
        f.write("RHS\n")
        if P["c0"] != 0:
            f.write(_line("", "RHS", obj, _num(-P["c0"])))     # objective constant = -(RHS of the objective row)
        for i in range(mr):
            if P["rhs"][i] != 0:
                f.write(_line("", "RHS", rn(i), _num(P["rhs"][i])))
        if P["ranges"]:
            f.write("RANGES\n")
            for i, r in sorted(P["ranges"].items()):
                f.write(_line("", "RNG", rn(i), _num(r)))
        if P["btype"]:
            f.write("BOUNDS\n")
            for j, (t, v) in sorted(P["btype"].items()):
                f.write(_line(t, "BND", cn(j), "" if v is None else _num(v)))
        f.write("QUADOBJ\n")
        for j in range(n):
            for i in range(j, n):
                if P["Q"][i, j] != 0:
                    f.write(_line("", cn(j), cn(i), _num(P["Q"][i, j])))
        f.write("ENDATA\n")


def main():
    os.makedirs(OUT, exist_ok=True)
    for k in range(COUNT):
        write(problem(k), os.path.join(OUT, "fix%02d.qps" % k))
    print("wrote %d files to %s" % (COUNT, OUT))


if __name__ == "__main__":
    main()
