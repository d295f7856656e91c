#!/usr/bin/env python3
"""CLI of qpalm_amd/asm_gate.py (the build gate against the ROCm 7.2 register-allocator fault): scan_exec_prologue.py file.s [-v]
-> the suspicious saves among the findings (all findings with -v), exit code 1 if any."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from qpalm_amd.asm_gate import copies, scan  # noqa: E402,F401

if __name__ == "__main__":
    found = scan(sys.argv[1])
    bad = copies(found)
    if "-v" in sys.argv:
        for fn, block, ln, s, outside in found:
            print("%s %s line %d: %s%s" % (fn, block, ln, s, "   <-- saves a value from before the region" if outside else ""))
    for fn, block, ln, s in bad:
        print("SAVE %s %s line %d: %s" % (fn, block, ln, s))
    print("%d vector instructions ahead of an exec restore, %d of them saves of outside values" % (len(found), len(bad)))
    sys.exit(1 if bad else 0)
