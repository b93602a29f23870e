"""Test infrastructure: lets the product's own Python binding (diffusiondepth_amd.backend.HipDenoiser, and through it head.py / modules.py)
drive the host-emulated library on CPU tensors, by patching the binding's PRIVATE guards from the outside (tensor-device check, stream lookup,
device scope, HipBound's device rule).  Nothing here is reachable from the product: it refuses CPU tensors and loads only libddepth_hip.so."""
from __future__ import annotations

import contextlib
import ctypes

import torch

from diffusiondepth_amd import backend as B_, modules as M_
from hostemu_util import build_library

CPU = torch.device("cpu")


def load():
    lib = build_library()
    for name, (res, args) in B_.abi_signatures().items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    lib.emu_launch_count.restype = ctypes.c_ulong
    return lib


def install(lib, setattr_fn):
    """setattr_fn(obj, name, value): monkeypatch.setattr in a test, plain setattr in a spawned worker.  Returns (factory, list of made backends)."""
    def check_tensor(t, name, shape=None, dtype=None):
        assert isinstance(t, torch.Tensor), name
        if dtype is not None and t.dtype != dtype:
            raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError(f"{name} must have shape {tuple(shape)}, got {tuple(t.shape)}")
        return t.contiguous()
    setattr_fn(B_, "_check_tensor", check_tensor)
    setattr_fn(B_, "_stream_ptr", lambda device: 0)
    setattr_fn(torch.cuda, "device", lambda d: contextlib.nullcontext())
    made = []

    def make(device, variant="res"):
        be = object.__new__(B_.HipDenoiser)
        be._lib, be.device, be.variant = lib, CPU, variant
        h = ctypes.c_void_p()
        assert lib.dd_create(ctypes.byref(h), 0, B_.VARIANTS[variant]) == 0
        be._h = h
        be._have_schedule = be._have_weights = be._have_fpn = be._have_neck = False
        be._cond_token = None
        made.append(be)
        return be
    setattr_fn(M_, "_on_hip_device", lambda t: True)      # CPU tensors take the (emulated) library, not the eager torch path
    setattr_fn(M_.HipBound, "_hip_device", staticmethod(lambda device: CPU))
    setattr_fn(M_.HipBound, "_make_backend", lambda self, device: make(device, self.variant))
    return make, made


def install_dcn(lib, setattr_fn):
    """Same for diffusiondepth_amd.dcn (the DCN-extension drop-in under nlspn.NLSPN): its tensor guard, stream lookup and library handle."""
    from diffusiondepth_amd import dcn

    def dev_f32(t, name):
        assert isinstance(t, torch.Tensor), name
        if t.dtype != torch.float32:
            raise RuntimeError(f"{name} must be float32, got {t.dtype}")
        if not t.is_contiguous():
            raise RuntimeError(f"{name} tensor has to be contiguous")
        return t
    setattr_fn(dcn, "_dev_f32", dev_f32)
    setattr_fn(dcn, "_stream", lambda t: ctypes.c_void_p(0))
    setattr_fn(dcn, "require_hip", lambda t, what: None)
    setattr_fn(torch.cuda, "device", lambda d: contextlib.nullcontext())
    setattr_fn(B_, "load_library", lambda: lib)
    setattr_fn(dcn, "_bound", None)                  # dcn._lib() then binds (and types) the emulated library's dd_dcn_* / dd_nlspn_* entries
    return dcn


def destroy(lib, made):
    for be in made:
        if be._h is not None:
            lib.dd_destroy(be._h)
            be._h = None
