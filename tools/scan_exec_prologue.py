#!/usr/bin/env python3
"""Scans gfx950 assembly (hipcc -save-temps *.s) for vector instructions that the compiler placed at the top of a basic
block BEFORE the `s_or_b64 exec, exec, ...` that re-enables the lanes of a finished divergent region.  Such an instruction
runs with the lanes of the region still switched off; when it is a register copy that carries a value across a later
call (live-range split), the lanes that were off keep garbage.  ROCm 7.2's register allocator does this (round 3:
info.dual_objective was wrong on the GPU because the lane-index register of one wave-shuffle step was saved this way
around a call to dense_solve).  Usage: scan_exec_prologue.py file.s [-v]  -> the register copies among the findings (all findings with -v), exit code 1 if any copy."""
import re, sys

def scan(path):
    fn, out, block, pending = None, [], None, []
    for ln, l in enumerate(open(path, errors="replace"), 1):
        s = l.strip()
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", l)
        if m and not l.startswith(".L"):
            fn = m.group(1)
        if re.match(r"^\.LBB\d+_\d+:", l):
            block, pending = s.split(":")[0], []
            continue
        if block is None or not s or s.startswith(";") or s.startswith("."):
            continue
        op = s.split()[0]
        if op == "s_or_b64" and s.replace(" ", "").startswith("s_or_b64exec,exec,"):
            for (pl, ps) in pending:
                out.append((fn, block, pl, ps))
            block = None
            continue
        if op.startswith(("v_", "ds_", "flat_", "global_", "scratch_", "buffer_")):
            pending.append((ln, s))
        elif op.startswith("s_") and op not in ("s_waitcnt", "s_nop", "s_mov_b64", "s_mov_b32", "s_barrier", "s_and_b64", "s_lshl_b32", "s_lshr_b64"):
            block = None  # control flow or another exec write: no longer the block prologue
        if len(pending) > 12:
            block = None
    return out

_COPY = re.compile(r"^(v_mov_b(32|64)_e32 v\[?[0-9:]+\]?, v\[?[0-9:]+\]?|v_accvgpr_(read|write)_b32 [av]\d+, [av]\d+)$")


def copies(found):
    """The findings that are plain register-to-register copies: a value that was live before the divergent region and is
    only being moved (a phi of the region is computed, not copied from a register that the region did not write)."""
    return [f for f in found if _COPY.match(f[3])]


if __name__ == "__main__":
    found = scan(sys.argv[1])
    bad = copies(found)
    if "-v" in sys.argv:
        for fn, block, ln, s in found:
            print("%s %s line %d: %s" % (fn, block, ln, s))
    for fn, block, ln, s in bad:
        print("COPY %s %s line %d: %s" % (fn, block, ln, s))
    print("%d vector instructions ahead of an exec restore, %d of them register copies" % (len(found), len(bad)))
    sys.exit(1 if bad else 0)
