"""The build gate against the ROCm 7.2 register-allocator fault ("compiler fault"): qpalm_amd/asm_gate.py (CLI: tools/evidence/scan_exec_prologue.py)
must flag a plain VGPR-to-VGPR copy that sits between a block label and the `s_or_b64 exec, exec, ...` of that block, and must not
flag computed values there (phis of the lanes that were active) nor copies behind the exec restore."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FAULTY = """
_ZN5qp5127k_solveILi2EEEv8qpg_viewii:
.LBB57_1093:
	flat_load_dwordx2 v[12:13], v[8:9]
	s_andn2_b64 exec, exec, s[2:3]
	s_cbranch_execnz .LBB57_1093
.LBB57_1094:                            ; %Flow2605
	v_mov_b64_e32 v[76:77], v[74:75]
	v_mov_b32_e32 v63, v113
	s_or_b64 exec, exec, s[0:1]
	s_barrier
	s_swappc_b64 s[30:31], s[0:1]
	v_mov_b32_e32 v113, v63
"""

CLEAN = """
_ZN5qp5127k_solveILi2EEEv8qpg_viewii:
.LBB57_1093:
	flat_load_dwordx2 v[12:13], v[8:9]
	s_andn2_b64 exec, exec, s[2:3]
	s_cbranch_execnz .LBB57_1093
.LBB57_1094:                            ; %Flow2605
	v_add_f64 v[12:13], v[12:13], v[14:15]
	ds_write_b64 v66, v[12:13] offset:17088
	s_or_b64 exec, exec, s[0:1]
	v_mov_b32_e32 v63, v113
	s_barrier
	s_swappc_b64 s[30:31], s[0:1]
	v_mov_b32_e32 v113, v63
"""


def _scan():
    from qpalm_amd import asm_gate
    return asm_gate


# legitimate: the copy moves a value the block itself computed (a phi of the lanes that were active), e.g. a counter bumped inside an
# `if (thread == 0)` region (round 4: the wave_rank assignment tripped the round-3 gate, which looked at the opcode only)
PHI_COPY = """
_ZN5qp5127k_solveILi2EEEv8qpg_viewii:
.LBB72_953:
	v_add_u32_e32 v0, 1, v8
	v_mov_b32_e32 v1, 0
	ds_write_b32 v1, v8 offset:1660
	v_mov_b32_e32 v8, v0
	s_or_b64 exec, exec, s[2:3]
	v_cmp_lt_u32_e32 vcc, 1, v2
"""

# other ways to save a pre-region value, and another form of the restore
SPILLS = """
_ZN5qp5127k_solveILi2EEEv8qpg_viewii:
.LBB57_2000:
	scratch_store_dword off, v113, s32 offset:16
	v_accvgpr_write_b32 a3, v77
	s_mov_b64 exec, s[4:5]
	s_barrier
.LBB57_2001:
	v_writelane_b32 v40, s30, 0
	s_or_saveexec_b64 s[6:7], s[8:9]
	s_barrier
"""


def test_gate_flags_copies_ahead_of_the_exec_restore(tmp_path):
    mod = _scan()
    bad, good = tmp_path / "bad.s", tmp_path / "good.s"
    bad.write_text(FAULTY)
    good.write_text(CLEAN)
    found = mod.scan(str(bad))
    assert [f[3] for f in mod.copies(found)] == ["v_mov_b64_e32 v[76:77], v[74:75]", "v_mov_b32_e32 v63, v113"]
    assert all(f[0].startswith("_ZN5qp5127k_solve") and f[1] == ".LBB57_1094" for f in found)
    found = mod.scan(str(good))
    assert len(found) == 2 and mod.copies(found) == []   # computed values ahead of the restore are legitimate; the copy sits behind it
    phi, sp = tmp_path / "phi.s", tmp_path / "spills.s"
    phi.write_text(PHI_COPY)
    sp.write_text(SPILLS)
    assert len(mod.scan(str(phi))) == 4 and mod.copies(mod.scan(str(phi))) == []
    # (v_writelane_b32 ignores EXEC: an SGPR spill into a VGPR lane is safe wherever it sits)
    assert [f[3].split()[0] for f in mod.copies(mod.scan(str(sp)))] == ["scratch_store_dword", "v_accvgpr_write_b32"]


# round 6: the ENTRY of a divergent region written as copy-of-exec / s_and / s_mov exec: exec is narrowed, the copy in front of it ran with
# the block's full mask -- not the fault (the loop preheaders of qp128::k_solve after the Newton-direction guard went in)
REGION_ENTRY = """
_ZN5qp1287k_solveILi1EEEv8qpg_viewii:
.LBB82_1016:                            ; %.preheader74.i
	s_mov_b64 s[0:1], exec
	v_readlane_b32 s2, v126, 54
	v_readlane_b32 s3, v126, 55
	s_and_b64 s[2:3], s[0:1], s[2:3]
	v_mov_b32_e32 v8, v124
	s_mov_b64 exec, s[2:3]
	s_cbranch_execz .LBB82_1019
"""


def test_gate_does_not_flag_the_entry_of_a_divergent_region(tmp_path):
    mod = _scan()
    f = tmp_path / "entry.s"
    f.write_text(REGION_ENTRY)
    assert mod.copies(mod.scan(str(f))) == []
    g = tmp_path / "restore.s"   # ... while the same copy ahead of a restore from a register the block did not derive from exec stays flagged
    g.write_text(REGION_ENTRY.replace("s_and_b64 s[2:3], s[0:1], s[2:3]", "s_nop 0"))
    assert [x[3] for x in mod.copies(mod.scan(str(g)))] == ["v_mov_b32_e32 v8, v124"]


REGION_BODY = """
_ZN5qp5128sp_solveEiRKNS_8SpArraysEPd:
.LBB9_110:                              ;   in Loop: Header=BB9_87 Depth=1
	v_mov_b32_e32 v13, v7
	v_lshl_add_u64 v[20:21], v[12:13], 2, v[14:15]
	flat_load_dword v68, v[20:21]
	s_or_b64 exec, exec, s[16:17]
	s_and_saveexec_b64 s[16:17], s[8:9]
	s_cbranch_execz .LBB9_109
"""


def test_gate_does_not_flag_a_copy_the_region_itself_consumes(tmp_path):
    """round 6 (pipelined sparse solves): the body of a one-block divergent region builds a 64-bit index from a loop-invariant zero and loads
    through it -- the copy is read before the restore, it is not a value parked for the lanes that are off; the same copy NOT read before the
    restore stays a finding"""
    mod = _scan()
    f = tmp_path / "body.s"
    f.write_text(REGION_BODY)
    assert mod.copies(mod.scan(str(f))) == []
    g = tmp_path / "parked.s"
    g.write_text(REGION_BODY.replace("v[12:13], 2", "v[30:31], 2"))
    assert [x[3] for x in mod.copies(mod.scan(str(g)))] == ["v_mov_b32_e32 v13, v7"]
