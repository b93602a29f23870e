"""Shared helpers of the host-emulation tests (tests/test_wino_host_emulation.py, tests/test_igemm2_host_emulation.py): building a kernel
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


def _compile(cxx, csrc_dir, so, unit="wino_host.cpp"):
    cmd = [cxx, "-std=c++17", "-O1", "-mf16c", "-x", "c++", "-DDD_HOST_EMULATION", "-Wno-psabi", "-Wno-unused-value", "-I", EMU, "-I", csrc_dir,
           "-shared", "-fPIC", os.path.join(EMU, unit), "-o", so + ".tmp"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.fail("host build of %s failed:\n%s" % (unit, r.stderr[-4000:]))
    os.replace(so + ".tmp", so)


def _bind(lib):
    P = ctypes.c_void_p
    lib.emu_wino_pack_bytes.restype = ctypes.c_longlong
    lib.emu_wino_pack.argtypes = [P, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
    lib.emu_wino_pack.restype = None
    lib.emu_wino_table.argtypes = [P, P, P, P, P] + [ctypes.c_int] * 6 + [P]
    lib.emu_wino_layer.argtypes = [ctypes.c_int] * 5 + [P] * 7 + [ctypes.c_int] * 3
    lib.emu_set_order.argtypes = [ctypes.c_int]
    lib.emu_set_dma_late.argtypes = [ctypes.c_int]
    return lib


def _build(unit="wino_host.cpp", kernel_sources=("dd_wino.hip",), env_override="DD_EMU_LIB"):
    """Compile tests/host_emul/<unit> (which #includes the kernel sources) into build/host_emul/, cached by the hash of everything it reads."""
    cxx = _clangxx()
    if cxx is None:
        pytest.skip("no clang++ (the kernels use clang vector extensions; g++ cannot compile them)")
    srcs = [os.path.join(EMU, unit), os.path.join(EMU, "hip", "hip_runtime.h")] + [os.path.join(CSRC, f) for f in kernel_sources] + \
           [os.path.join(CSRC, f) for f in ("dd_elem.h", "dd_kernels.h", "dd_gcn.h", "dd_igemm2_cfg.h")]
    hsh = hashlib.sha1()
    for s in srcs:
        with open(s, "rb") as f:
            hsh.update(f.read())
    if os.environ.get(env_override):            # a hand-built variant (mutation experiments)
        return ctypes.CDLL(os.environ[env_override])
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, "lib%s_%s.so" % (unit.split(".")[0], hsh.hexdigest()[:12]))
    if not os.path.exists(so):
        _compile(cxx, CSRC, so, unit)
    return ctypes.CDLL(so)


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


