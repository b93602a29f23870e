"""Shared helpers of the host-emulation tests (tests/test_igemm2_host_emulation.py, tests/test_library_host_emulation.py): building a kernel
translation unit for the host on top of tests/host_emul/hip/hip_runtime.h, the 16-bit element kinds and the activation layouts."""
from __future__ import annotations

import ctypes
import hashlib
import os
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "diffusiondepth_amd", "csrc")
EMU = os.path.join(ROOT, "tests", "host_emul")
OUT = os.path.join(ROOT, "build", "host_emul")
EK_BF16, EK_F16 = 1, 2
STAT_SLOTS, STAT_STRIDE, GN_GROUPS = 32, 16, 4


def _clangxx():
    for c in ("/opt/rocm/lib/llvm/bin/clang++", shutil.which("clang++")):
        if c and os.path.exists(c):
            return c
    return None


def have_f16c():
    return os.path.exists("/proc/cpuinfo") and "f16c" in open("/proc/cpuinfo").read()


# ---- 16-bit element kinds and the channel-blocked activation layout of dd_elem.h ([B][C/32][h][w][32]) ---------------------------------------
def to16(x, ek):
    x = np.ascontiguousarray(x, dtype=np.float32)
    if ek == EK_F16:
        return x.astype(np.float16).view(np.uint16)
    u = x.view(np.uint32).astype(np.uint64)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)        # round to nearest even


def from16(u, ek):
    if ek == EK_F16:
        return u.view(np.float16).astype(np.float32)
    return (u.astype(np.uint32) << 16).view(np.float32)


def blocked(x_nchw16):
    B, C, h, w = x_nchw16.shape
    return np.ascontiguousarray(x_nchw16.reshape(B, C // 32, 32, h, w).transpose(0, 1, 3, 4, 2))


def unblocked(x_blk, C):
    B, nb, h, w, _ = x_blk.shape
    return np.ascontiguousarray(x_blk.transpose(0, 1, 4, 2, 3).reshape(B, C, h, w))


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None




# ---- the whole library for the host: every csrc source + the harness units of tests/host_emul ------------------------------------------------------
LIB_SOURCES = ("dd_api.cpp", "dd_api_weights.cpp", "dd_api_plans.cpp", "dd_api_train.cpp", "dd_igemm2.hip", "dd_misc.hip", "dd_naive.hip", "dd_bwd.hip", "dd_wgrad.hip", "dd_wgrad2.hip", "dd_dcn.hip", "dd_thin.hip", "dd_msda.hip")
HARNESS_UNITS = ("ddepth_host.cpp", "igemm2_host.cpp")
_FLAGS = ["-std=c++17", "-O1", "-mf16c", "-x", "c++", "-DDD_HOST_EMULATION", "-Wno-psabi", "-Wno-unused-value", "-fPIC", "-c"]


def _extra_flags():
    """DD_EMU_CFLAGS="-DDD_DMA_SPREAD=1 ...": the kernels' compile-time options (dd_igemm2_cfg.h) in the emulated build -- how a variant is
    checked under the adversarial schedules before it meets the GPU"""
    return os.environ.get("DD_EMU_CFLAGS", "").split()


def _cc(cxx, src, obj, incs):
    cmd = [cxx] + _FLAGS + _extra_flags() + [a for i in incs for a in ("-I", i)] + [src, "-o", obj]
    return subprocess.run(cmd, capture_output=True, text=True)


class _BuildLock:
    """One host build at a time across processes (pytest-xdist workers share build/host_emul): the first worker compiles, the others find its cache."""

    def __enter__(self):
        import fcntl
        os.makedirs(OUT, exist_ok=True)
        self.f = open(os.path.join(OUT, ".lock"), "w")
        fcntl.flock(self.f, fcntl.LOCK_EX)

    def __exit__(self, *a):
        self.f.close()


def _objects():
    """Compile (in parallel, cached by the hash of every input) all translation units; -> (cxx, {unit: object path}).  Call under _BuildLock."""
    from concurrent.futures import ThreadPoolExecutor
    cxx = _clangxx()
    if cxx is None:
        pytest.skip("no clang++ (the kernels use clang vector extensions; g++ cannot compile them)")
    if not have_f16c():
        pytest.skip("host without F16C")
    hsh = hashlib.sha1()
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".cpp")))
    deps += [os.path.join(EMU, u) for u in HARNESS_UNITS] + [os.path.join(EMU, "hip", "hip_runtime.h"),
                                                              os.path.join(ROOT, "include", "ddepth.h"), os.path.join(ROOT, "include", "ddepth_dcn.h"),
                                                              os.path.join(ROOT, "include", "ddepth_msda.h")]
    for s in deps:
        with open(s, "rb") as f:
            hsh.update(f.read())
    hsh.update(" ".join(_extra_flags()).encode())
    objdir = os.path.join(OUT, "obj_" + hsh.hexdigest()[:12])
    if not os.path.isdir(objdir) and os.path.isdir(OUT) and not _extra_flags():          # a new source state: drop the builds of older ones
        for d in os.listdir(OUT):
            if d.startswith("obj_"):
                shutil.rmtree(os.path.join(OUT, d), ignore_errors=True)
    os.makedirs(objdir, exist_ok=True)
    units = [os.path.join(CSRC, f) for f in LIB_SOURCES] + [os.path.join(EMU, u) for u in HARNESS_UNITS]
    objs = {os.path.basename(u): os.path.join(objdir, os.path.basename(u).rsplit(".", 1)[0] + ".o") for u in units}
    todo = [u for u in units if not os.path.exists(objs[os.path.basename(u)])]
    if todo:
        def cc(src):
            tmp = objs[os.path.basename(src)] + ".tmp.o"
            r = _cc(cxx, src, tmp, [EMU, CSRC])
            if r.returncode == 0:
                os.replace(tmp, objs[os.path.basename(src)])
            return src, r
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
            for src, r in ex.map(cc, todo):
                if r.returncode != 0:
                    pytest.fail("host build of %s failed:\n%s" % (src, r.stderr[-4000:]))
    return cxx, objs, objdir


def _link(cxx, objs, so):
    r = subprocess.run([cxx, "-shared", "-fPIC", "-o", so + ".tmp"] + list(objs), capture_output=True, text=True)
    if r.returncode != 0:
        pytest.fail("host link failed:\n" + r.stderr[-4000:])
    os.replace(so + ".tmp", so)


def build_library():
    """build/host_emul/obj_<hash>/libddepth_hostemu.so: the complete C ABI (include/ddepth.h, include/ddepth_dcn.h) plus the harness entry points
    (emu_*) executing on the CPU.  DD_EMU_LIB=<path> substitutes a hand-built variant."""
    if os.environ.get("DD_EMU_LIB"):
        return ctypes.CDLL(os.environ["DD_EMU_LIB"])
    with _BuildLock():
        cxx, objs, objdir = _objects()
        so = os.path.join(objdir, "libddepth_hostemu.so")
        if not os.path.exists(so):
            _link(cxx, objs.values(), so)
    return ctypes.CDLL(so)


def build_mutant(unit, old, new, tmp_path, count=None):
    """The library with ONE source file textually changed (old -> new): only that unit is recompiled.  For the tests that check that the
    emulation notices a broken kernel."""
    with _BuildLock():
        cxx, objs, _ = _objects()
    src = open(os.path.join(CSRC, unit)).read()
    n = src.count(old)
    assert n >= 1 and (count is None or n == count), "mutation anchor not found %d times in %s: the source changed, update the test" % (n, unit)
    d = os.path.join(str(tmp_path), "csrc")
    os.makedirs(d, exist_ok=True)
    mut = os.path.join(d, unit)
    with open(mut, "w") as f:
        f.write(src.replace(old, new))
    obj = os.path.join(str(tmp_path), "mut.o")
    r = _cc(cxx, mut, obj, [EMU, CSRC])            # the unchanged headers come from the real csrc directory
    if r.returncode != 0:
        pytest.fail("host build of the mutated %s failed:\n%s" % (unit, r.stderr[-4000:]))
    so = os.path.join(str(tmp_path), "libmut.so")
    _link(cxx, [obj if k == unit else o for k, o in objs.items()], so)
    return ctypes.CDLL(so)


def bind_igemm2(lib):
    P = ctypes.c_void_p
    lib.emu_set_order.argtypes = [ctypes.c_int]
    lib.emu_set_dma_late.argtypes = [ctypes.c_int]
    lib.emu_geom2.argtypes = [ctypes.c_int, ctypes.c_int, P]
    lib.emu_conv2.argtypes = [ctypes.c_int, ctypes.c_int] + [P] * 11 + [ctypes.c_int] * 2 + [P] * 3 + [ctypes.c_int] * 4 + [P] * 3
    return lib
