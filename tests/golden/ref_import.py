"""Test-only import shim for the REAL reference (read-only at /root/reference).

Only usable in the build container: it loads the reference's own Python classes so that
``make_golden.py`` can mint known-answer vectors from them.  Nothing here is copied from the
reference; the shim merely supplies stand-ins for the *container / factory* names of mmcv and
mmdet3d that the reference head files import (those packages are not installed here).  Every
arithmetic operation executed afterwards is the reference's own code plus stock ``torch.nn``.

Stand-ins and the reference lines that need them:
  mmcv.runner.BaseModule/ModuleList/force_fp32   src/model/head/ddim_depth_estimate_res.py:7
  mmcv.cnn.ConvModule/build_*_layer              src/model/head/ddim_depth_estimate_res.py:8,
                                                 src/model/head/ddim_depth_estimate_res_swin_add.py (ConvModule)
  mmcv.utils.Registry                            src/model/ops/depth_transform.py:1,7
  mmdet3d.models.builder.HEADS/build_loss        src/model/head/ddim_depth_estimate_res.py:6,14
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types

import torch
from torch import nn

REF_ROOT = os.environ.get("DD_REFERENCE_ROOT", "/root/reference")
REF_SRC = os.path.join(REF_ROOT, "src")
# Where /root/reference does not exist (the GPU box): the same modules as sourceless bytecode, compiled from the reference where it lies by
# oracle/ref_py/build_ref.py into the git-ignored oracle/_ref/py/ (what bench.py's cpu_baseline of kind "reference" times)
STAGED_SRC = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle", "_ref", "py")
if not os.path.isdir(os.path.join(REF_SRC, "model", "head")) and os.path.isdir(os.path.join(STAGED_SRC, "model", "head")):
    REF_SRC = STAGED_SRC


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_SRC, "model", "head"))


def reference_kind() -> str:
    return "staged bytecode (oracle/_ref/py)" if REF_SRC == STAGED_SRC else "source tree"


class _Registry:
    def __init__(self, name):
        self.name = name
        self._mods = {}

    def register_module(self, name=None, force=False, module=None):
        def deco(cls):
            self._mods[name or cls.__name__] = cls
            return cls
        return deco

    def build(self, cfg):
        cfg = dict(cfg)
        return self._mods[cfg.pop("type")](**cfg)

    def get(self, key):
        return self._mods.get(key)


class _BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg

    def init_weights(self):
        pass


class _ConvModule(nn.Module):
    """mmcv ConvModule with norm_cfg=None, act_cfg=None == Conv2d(+bias) stored under ``.conv``."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0,
                 conv_cfg=None, norm_cfg=None, act_cfg=None, **kw):
        super().__init__()
        assert norm_cfg is None and act_cfg is None, "stub only covers the plain-conv use in the reference"
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=True)

    def forward(self, x):
        return self.conv(x)


def _build_norm_layer(cfg, num_features):
    assert cfg["type"] == "BN"
    return "bn", nn.BatchNorm2d(num_features)


def _build_upsample_layer(cfg, in_channels, out_channels, kernel_size, stride, **kw):
    assert cfg["type"] == "deconv"
    return nn.ConvTranspose2d(in_channels, out_channels, kernel_size, stride, bias=cfg.get("bias", True))


def _build_conv_layer(cfg, *a, **kw):
    return nn.Conv2d(*a, **kw)


def _force_fp32(*a, **kw):
    def deco(fn):
        return fn
    return deco


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    if "mmcv" in sys.modules and getattr(sys.modules["mmcv"], "_dd_stub", False):
        return
    mod("mmcv", _dd_stub=True)
    mod("mmcv.runner", BaseModule=_BaseModule, ModuleList=nn.ModuleList, force_fp32=_force_fp32)
    mod("mmcv.cnn", ConvModule=_ConvModule, build_conv_layer=_build_conv_layer,
        build_norm_layer=_build_norm_layer, build_upsample_layer=_build_upsample_layer)
    mod("mmcv.utils", Registry=_Registry)
    if "torchvision" not in sys.modules:
        try:
            importlib.import_module("torchvision")
        except ImportError:   # only named by NLSPN's resnet factories (reference src/model/common.py:18,26-43)
            mod("torchvision")
    mod("mmdet3d")
    mod("mmdet3d.models")
    mod("mmdet3d.models.builder", HEADS=_Registry("heads"), build_loss=lambda cfg: None)


def _load_as(modname, relpath):
    path = os.path.join(REF_SRC, relpath)
    if not os.path.exists(path):
        path = path[:-3] + ".pyc"          # staged bytecode
    spec = importlib.util.spec_from_file_location(modname, path)
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


_CACHE = {}


def load_reference():
    """Returns a namespace with the reference's own classes (imported verbatim from /root/reference)."""
    if _CACHE:
        return _CACHE["ns"]
    if not reference_available():
        raise RuntimeError(f"reference tree not found under {REF_ROOT}")
    _install_stubs()
    if REF_SRC not in sys.path:
        sys.path.insert(0, REF_SRC)
    # `model` and `model.diffusers*` / `model.ops.depth_transform` import as-is; `model.head` and
    # `model.ops` get synthetic package objects so their __init__ (which drags in mmcv.ops / numba /
    # cv2) is skipped.
    importlib.import_module("model")
    for pkg in ("model.head", "model.ops"):
        p = types.ModuleType(pkg)
        p.__path__ = [os.path.join(REF_SRC, *pkg.split("."))]
        sys.modules[pkg] = p
    # depth_map_to_points is imported by the base head but unused on the DDIM path
    sys.modules["model.ops.depth_map_to_points"] = types.ModuleType("model.ops.depth_map_to_points")
    sys.modules["model.ops.depth_map_to_points"].convert_depth_map_to_points = None
    sched = importlib.import_module("model.diffusers.schedulers.scheduling_ddim")
    dt = _load_as("model.ops.depth_transform", "model/ops/depth_transform.py")
    _load_as("model.head.mmbev_base_depth_refine", "model/head/mmbev_base_depth_refine.py")
    res = _load_as("model.head.ddim_depth_estimate_res", "model/head/ddim_depth_estimate_res.py")
    swin = _load_as("model.head.ddim_depth_estimate_res_swin_add", "model/head/ddim_depth_estimate_res_swin_add.py")
    ns = types.SimpleNamespace(
        DDIMScheduler=sched.DDIMScheduler,
        DeepDepthTransformWithUpsampling=dt.DeepDepthTransformWithUpsampling,
        ScheduledCNNRefine=res.ScheduledCNNRefine,
        CNNDDIMPipiline=res.CNNDDIMPipiline,
        DDIMDepthEstimate_Res=res.DDIMDepthEstimate_Res,
        ScheduledCNNRefineSwin=swin.ScheduledCNNRefine,
        CNNDDIMPipilineSwin=swin.CNNDDIMPipiline,
        DDIMDepthEstimate_Swin_ADD=swin.DDIMDepthEstimate_Swin_ADD,
    )
    _CACHE["ns"] = ns
    return ns


def load_weights(module: nn.Module, sd: dict, prefix: str):
    """Load the ``prefix``-ed subset of a synth state dict (numpy arrays) into a reference module."""
    own = module.state_dict()
    sub = {k[len(prefix):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith(prefix)}
    for k in own:
        if k.endswith("num_batches_tracked"):
            sub[k] = own[k]
    missing = set(own) - set(sub)
    extra = set(sub) - set(own)
    assert not missing and not extra, (missing, extra)
    module.load_state_dict(sub, strict=True)
    return module


# ---- the reference's own model FACADE (src/model/diffusion_dcbase_model.py:26-224), for the drop-in test tests/test_reference_facade.py ----------
_FACADE = {}


def load_reference_facade():
    """The reference's unmodified ``Diffusion_DCbase_Model`` class, imported from the reference tree (or its staged bytecode) behind stand-ins for the
    CONTAINER names its module imports at the top (diffusion_dcbase_model.py:11-21):
      mmdet.models.DETECTORS / BACKBONES, mmdet3d.models.builder, mmdet3d.core.bbox3d2result, mmdet3d.utils.collect_env / get_root_logger
                                  registries and helpers the class only names (a decorator and unused imports)
      model.ops.ip_basic          classical depth completion (cv2): reached only with ip_basic=True, which no configuration sets
      model.backbone.get          the reference's rule `getattr(import_module('model.backbone.' + module), name)` needs mmdet's ResNet blocks; the stand-in
                                  serves the zero-argument factories the TEST registers in ``BACKBONE_FACTORIES`` (the same backbone on both sides of a comparison)
      model.head.get              named, never called (the head comes from HEADS.build, :91)
    Returns the loaded module; its global ``HEADS`` is the registry ``Diffusion_DCbase_Model.__init__`` builds the head from (:91) -- the test rebinds it to a
    registry holding either the reference's head classes or diffusiondepth_amd's."""
    if _FACADE:
        return _FACADE["mod"]
    load_reference()
    m = sys.modules

    def mod(name, **attrs):
        mm = m.get(name) or types.ModuleType(name)
        mm.__dict__.update(attrs)
        m[name] = mm
        return mm
    mod("mmdet")
    mod("mmdet.models", DETECTORS=_Registry("detectors"), BACKBONES=_Registry("backbones"))
    mod("mmdet3d.models", builder=m["mmdet3d.models.builder"])
    mod("mmdet3d.core", bbox3d2result=None)
    mod("mmdet3d.utils", collect_env=None, get_root_logger=None)
    mod("model.ops.ip_basic")
    m["model.ops"].ip_basic = m["model.ops.ip_basic"]
    bb = mod("model.backbone", BACKBONE_FACTORIES={})
    bb.get = lambda args: bb.BACKBONE_FACTORIES[args.backbone_name]
    m["model.head"].get = lambda args: None
    fac = _load_as("model.diffusion_dcbase_model", "model/diffusion_dcbase_model.py")
    _FACADE["mod"] = fac
    return fac


def new_registry(name="heads"):
    return _Registry(name)
