"""Build helpers.  The product library is hipcc-only (gfx950); the emulation build is test-only."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libqpalm_gfx950.so")
EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_LIB = os.path.join(EMU_DIR, "libqpalm_gfx950_emu.so")
_SRCS = ["../host/qpalm_host.c", "../host/qpalm_qps.c", "qpalm_kkt.h", "qpalm_sparse.h", "qpalm_gfx950.hip", "qpalm_kernels.h", "qpalm_device.h", "qpalm_dense.h", "qpalm_iter.h", "qpalm_types.h",
         "qpalm_capi.inc"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _deps():
    return [os.path.join(CSRC, s) for s in _SRCS] + [os.path.join(ROOT, "include", "qpalm_gfx950.h")]


def build_hip(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    if not force and not _stale(LIB, _deps()):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # -enable-ipra=0: with LLVM's inter-procedural register allocation a kernel that calls two device
    # functions faulted on the GPU (clean with one callee, in the emulator and fully inlined), and every
    # edit inside a callee re-allocated the caller; the standard calling convention keeps phases independent.
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
           "-mllvm", "-enable-ipra=0",
           "-Wno-unused-value", "-o", LIB, os.path.join(CSRC, "qpalm_gfx950.hip")]
    cmd += os.environ.get("QPALM_EXTRA_DEFS", "").split()  # experiments: e.g. -DQP_UHELP=1
    if verbose:
        print(" ".join(cmd))
    # -save-temps into a scratch directory: the device assembly is checked for a ROCm 7.2 register-allocator fault
    # (tools/evidence/scan_exec_prologue.py: register copies ahead of an exec restore; round 3's wrong dual objective on the GPU)
    import shutil
    import tempfile
    tmp = tempfile.mkdtemp(prefix="qpalm_build_")
    try:
        subprocess.check_call(cmd + ["-save-temps"], cwd=tmp)
        asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")]
        if not asm:
            raise RuntimeError("hipcc -save-temps left no gfx950 assembly in %s" % tmp)
        check_device_assembly(os.path.join(tmp, asm[0]))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    build_host()
    return LIB


def check_device_assembly(path):
    """Fails the build when the device assembly saves a pre-region value ahead of the exec restore of its block (qpalm_amd/asm_gate.py).
    The library that was just linked is removed: a build that trips the gate must not be loadable."""
    from . import asm_gate
    try:
        asm_gate.check(path)
    except RuntimeError:
        if os.path.exists(LIB):
            os.remove(LIB)
        raise


def build_host():
    """The C host library (reference API names, include/qpalm_host.h) on top of the HIP library."""
    out = os.path.join(HERE, "lib", "libqpalm.so")
    libdir = os.path.dirname(LIB)
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-fPIC", "-shared", "-Wall", "-o", out,
                           os.path.join(HERE, "host", "qpalm_host.c"), os.path.join(HERE, "host", "qpalm_qps.c"),
                           "-L" + libdir, "-lqpalm_gfx950", "-Wl,-rpath,$ORIGIN", "-lm"])
    # the reference's CLI (interfaces/qps/src/qpalm_qps.c main): qpalm_qps problem.qps [settings.txt]
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-DQPALM_QPS_MAIN", "-Wall", "-o", os.path.join(HERE, "lib", "qpalm_qps"),
                           os.path.join(HERE, "host", "qpalm_qps.c"), "-L" + libdir, "-lqpalm", "-lqpalm_gfx950", "-Wl,-rpath,$ORIGIN", "-lm"])
    return out


def build_host_emu():
    """TEST-ONLY: the same host C layer on top of the emulated kernels (tests/emu)."""
    out = os.path.join(EMU_DIR, "libqpalm_host_emu.so")
    build_emu()
    subprocess.check_call(["gcc", "-O2", "-std=c99", "-D_POSIX_C_SOURCE=200809L", "-fPIC", "-shared", "-Wall", "-o", out,
                           os.path.join(HERE, "host", "qpalm_host.c"), os.path.join(HERE, "host", "qpalm_qps.c"),
                           "-L" + EMU_DIR, "-lqpalm_gfx950_emu", "-Wl,-rpath," + EMU_DIR, "-lm"])
    return out


def build_emu(force=False, block=128):
    """TEST-ONLY: the same kernel source compiled for the host against tests/emu/hip_emu.h."""
    deps = _deps() + [os.path.join(EMU_DIR, f) for f in ("hip_emu.h", "hip_emu.cpp", "qpalm_emu.cpp")]
    if not force and not _stale(EMU_LIB, deps):
        return EMU_LIB
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-DQP_T=%d" % block,
           "-Wno-unknown-pragmas", os.path.join(EMU_DIR, "qpalm_emu.cpp"), os.path.join(EMU_DIR, "hip_emu.cpp"),
           "-o", EMU_LIB]
    cmd += os.environ.get("QPALM_EXTRA_DEFS", "").split()
    subprocess.check_call(cmd)
    return EMU_LIB


def build_oracle(force=False):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"] + (["-B"] if force else []))
    return os.path.join(ROOT, "oracle", "libqpalm_oracle.so")
