"""Build gate against a register-allocator fault of ROCm 7.2 ("compiler fault").

Scans gfx950 assembly (hipcc -save-temps *.s) for vector instructions that the compiler placed at the top of a basic block BEFORE the
instruction that re-enables the lanes of a finished divergent region (`s_or_b64 exec, exec, ...`, `s_mov_b64 exec, ...`,
`s_or_saveexec_b64 ...`).  Such an instruction runs with the lanes of the region still switched off.  That is legitimate for values the
region itself computes (phis of the lanes that were active).  It is the fault when the instruction only SAVES a value that was live
before the region -- a register copy, an AGPR move or a spill to scratch whose source the block did not write: the
lanes that were off keep garbage in the saved copy (round 3: info.dual_objective was wrong on the GPU because the lane-index register
of one wave-shuffle step was saved this way around a call to dense_solve)."""
import re

_EXEC_RESTORE = re.compile(r"^(s_or_b64\s+exec,\s*exec,|s_mov_b64\s+exec,|s_or_saveexec_b64\b)")
_PASS_THROUGH = ("s_waitcnt", "s_nop", "s_mov_b64", "s_mov_b32", "s_barrier", "s_and_b64", "s_lshl_b32", "s_lshr_b64")
_REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")


def _regs(text):
    """set of (file, index) of the VGPRs / AGPRs named in an operand string"""
    out = set()
    for m in _REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), k) for k in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def _dst_src(ins):
    """(registers written, registers read) of a vector instruction, by operand position (first operand = destination, except stores)"""
    op, _, rest = ins.partition(" ")
    ops = [o.strip() for o in rest.split(",")]
    if not ops or not ops[0]:
        return set(), set()
    if op.startswith(("scratch_store", "global_store", "flat_store", "buffer_store", "ds_write")):
        return set(), _regs(rest)
    return _regs(ops[0]), _regs(",".join(ops[1:]))


# (v_writelane_b32 -- how SGPRs are spilled into VGPR lanes -- is NOT in the list: it ignores EXEC and writes its lane whatever the mask)
_SAVE = re.compile(r"^(v_mov_b(32|64)_e32\s+v\[?[0-9:]+\]?,\s*v\[?[0-9:]+\]?$|v_accvgpr_(read|write)_b32\s+[av]\d+,\s*[av]\d+$|scratch_store_\w+\s)")


_EXEC_COPY = re.compile(r"^s_mov_b64\s+(s\[\d+:\d+\]),\s*exec$")
_SAND = re.compile(r"^s_and_b64\s+(s\[\d+:\d+\]),\s*(s\[\d+:\d+\]|exec|vcc),\s*(s\[\d+:\d+\]|exec|vcc)$")
_EXEC_FROM = re.compile(r"^s_mov_b64\s+exec,\s*(s\[\d+:\d+\])$")


def scan(path):
    """[(function, block label, line, instruction, saves_outside_value)] for every vector instruction between a block label and the
    exec restore of that block"""
    fn, out, block, pending, written = None, [], None, [], set()
    exec_copies, narrowed = set(), set()
    for ln, l in enumerate(open(path, errors="replace"), 1):
        s = l.strip()
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", l)
        if m and not l.startswith(".L"):
            fn = m.group(1)
        if re.match(r"^\.LBB\d+_\d+:", l):
            block, pending, written = s.split(":")[0], [], set()
            exec_copies, narrowed = set(), set()
            continue
        if block is None or not s or s.startswith(";") or s.startswith("."):
            continue
        op = s.split()[0]
        # `s_mov_b64 s[a:b], exec` ... `s_and_b64 s[c:d], s[a:b], cond` ... `s_mov_b64 exec, s[c:d]` is the ENTRY of a divergent region: exec is narrowed to
        # a subset of what it is in this block, whatever stands in front of it ran with the block's full mask (round 6: the loop preheaders of
        # qp128::k_solve took this shape and the gate, which treated every write of exec as a restore, called their copies of `n` a fault)
        m1 = _EXEC_COPY.match(s)
        if m1:
            exec_copies.add(m1.group(1))
        m2 = _SAND.match(s)
        if m2 and ((m2.group(2) in exec_copies or m2.group(2) == "exec") or (m2.group(3) in exec_copies or m2.group(3) == "exec")):
            narrowed.add(m2.group(1))
        m3 = _EXEC_FROM.match(s)
        if m3 and m3.group(1) in narrowed:
            block = None
            continue
        if _EXEC_RESTORE.match(s):
            out.extend((fn, block, pl, ps, outside) for (pl, ps, outside, _pd) in pending)
            block = None
            continue
        if op.startswith(("v_", "ds_", "flat_", "global_", "scratch_", "buffer_")):
            dst, src = _dst_src(s)
            # a pure save (copy / AGPR move / spill / writelane) of a register this block has not written = a value from before the region
            outside = bool(_SAVE.match(s)) and not (src & written) and bool(src)
            # a copy that a later instruction of the SAME block reads before the restore is the region's own arithmetic (round 6: the body of
            # a one-block region builds a 64-bit index from a loop-invariant zero, `v_mov v13, v7; v_lshl_add_u64 .., v[12:13], ..; flat_load`),
            # not a value parked for the lanes that are off: the fault's copies sit in front of the restore for use behind it
            for k, (pl, ps, po, pd) in enumerate(pending):
                if po and (pd & src):
                    pending[k] = (pl, ps, False, pd)
            pending.append((ln, s, outside, dst))
            written |= dst
        elif op.startswith("s_") and op not in _PASS_THROUGH:
            block = None  # control flow or another exec write: no longer the block prologue
        if len(pending) > 12:
            block = None
    return out


def copies(found):
    """the findings that only save a value from before the divergent region: the fault"""
    return [f[:4] for f in found if f[4]]


def check(path):
    """raises RuntimeError when the assembly holds the pattern"""
    bad = copies(scan(path))
    if bad:
        raise RuntimeError("register saves ahead of an exec restore (compiler fault, see qpalm_amd/asm_gate.py):\n" +
                           "\n".join("%s %s line %d: %s" % b for b in bad[:20]))
