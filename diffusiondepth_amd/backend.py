"""ctypes binding of the C ABI in include/ddepth.h (the reference-side binding a maintainer would add;
see INTEGRATION.md).  PyTorch is used only for device memory and the current HIP stream."""
from __future__ import annotations

import ctypes
import os
from typing import Dict, Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_NAME = "libddepth_hip.so"
_lib = None

# "f16x3": split f16 (DD_PREC_F16X3, include/ddepth.h) -- the abs-1e-3-on-depth mode; its backward (round 6) sends f16 gradients behind the split forward
# "f16r": refined f16 (DD_PREC_F16R) -- f16 operands with one MFMA per product on the two large convolutions, the thin layers / the once-per-image
# term / the y3 hand-over made (near-)exact: the 16-bit inference mode that holds 1e-3 depth RMSE at KITTI's depth range (Res denoiser, forward only)
PRECISIONS = {"naive_fp32": 0, "fp32": 1, "bf16": 2, "f16": 3, "fp16": 3, "f16x3": 4, "split_f16": 4, "f16r": 5, "refined_f16": 5}
VARIANTS = {"res": 0, "swin": 1}

# every symbol include/ddepth.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = [
    "dd_create", "dd_destroy", "dd_last_error", "dd_version", "dd_set_weight", "dd_set_weight_device", "dd_commit_weights",
    "dd_set_schedule", "dd_condition", "dd_neck_condition", "dd_denoise", "dd_denoise_trace", "dd_denoise_once", "dd_denoise_once_backward", "dd_denoise_backward", "dd_zero_grad", "dd_get_grad", "dd_add_noise", "dd_encode", "dd_decode",
    "dd_set_option", "dd_last_loop_ms", "dd_get_counter", "dd_get_layer_ms", "dd_debug_fetch", "dd_debug_weights_digest",
]


def precision_id(p) -> int:
    if isinstance(p, int):
        return p
    try:
        return PRECISIONS[str(p).lower()]
    except KeyError:
        raise ValueError(f"unknown precision {p!r}; choose one of {sorted(PRECISIONS)}") from None


def library_path() -> str:
    return os.environ.get("DDEPTH_LIBRARY", os.path.join(_HERE, _LIB_NAME))


def abi_signatures():
    """ctypes (restype, argtypes) of every entry point of include/ddepth.h this module binds."""
    c_int, c_i64, c_vp, c_cp = ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_char_p
    fp = ctypes.POINTER(ctypes.c_float)
    return {
        "dd_create": (c_int, [ctypes.POINTER(c_vp), c_int, c_int]),
        "dd_destroy": (c_int, [c_vp]),
        "dd_last_error": (c_cp, [c_vp]),
        "dd_version": (c_cp, []),
        "dd_set_weight": (c_int, [c_vp, c_cp, c_vp, c_i64]),
        "dd_set_weight_device": (c_int, [c_vp, c_cp, c_vp, c_i64, c_vp]),
        "dd_commit_weights": (c_int, [c_vp, c_vp]),
        "dd_set_schedule": (c_int, [c_vp, c_vp, c_int]),
        "dd_condition": (c_int, [c_vp, ctypes.POINTER(c_vp), ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_int, c_int, c_vp, c_int, c_vp]),
        "dd_neck_condition": (c_int, [c_vp, ctypes.POINTER(c_vp), ctypes.POINTER(c_int), ctypes.POINTER(c_int), c_int, c_int, c_vp, c_int, c_vp]),
        "dd_denoise": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
        "dd_denoise_trace": (c_int, [c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
        "dd_denoise_once": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
        "dd_denoise_once_backward": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
        "dd_denoise_backward": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_vp]),
        "dd_zero_grad": (c_int, [c_vp, c_vp]),
        "dd_get_grad": (c_int, [c_vp, c_cp, c_vp, c_i64, c_vp]),
        "dd_add_noise": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_int, c_int, c_int, c_vp]),
        "dd_encode": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
        "dd_decode": (c_int, [c_vp, c_vp, c_vp, c_int, c_int, c_int, c_vp]),
        "dd_set_option": (c_int, [c_vp, c_cp, c_i64]),
        "dd_last_loop_ms": (c_int, [c_vp, fp]),
        "dd_get_counter": (c_int, [c_vp, c_cp, ctypes.POINTER(c_i64)]),
        "dd_get_layer_ms": (c_int, [c_vp, c_int, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64)]),
        "dd_debug_fetch": (c_int, [c_vp, c_cp, c_vp, c_i64, c_vp]),
        "dd_debug_weights_digest": (c_int, [c_vp, ctypes.POINTER(ctypes.c_uint64)]),
    }


def load_library():
    """Loads libddepth_hip.so.  Fails loudly: there is no eager / CPU fallback for the hot path."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build it with `python -m diffusiondepth_amd.build` (hipcc, gfx950). "
            "diffusiondepth_amd has no fallback path for the DDIM hot loop.")
    lib = ctypes.CDLL(path)
    sig = abi_signatures()
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def _torch():
    import torch
    return torch


def _stream_ptr(device) -> int:
    torch = _torch()
    return int(torch.cuda.current_stream(device).cuda_stream)


def _check_tensor(t, name, shape=None, dtype=None):
    torch = _torch()
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: the DDIM hot path runs only on a HIP device "
                           "(diffusiondepth_amd has no CPU fallback)")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} must have shape {tuple(shape)}, got {tuple(t.shape)}")
    return t.contiguous()


class HipDenoiser:
    """Owner of one dd_handle_t (one per device/process).  All tensor arguments are torch CUDA
    (= HIP) tensors, fp32 NCHW, exactly what the reference modules exchange."""

    last_trajectory_ticket = 0      # ticket of the last denoise(..., keep_trajectory=True) call (0 = none)

    def __init__(self, device=None, variant: str = "res"):
        torch = _torch()
        self._lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible to PyTorch: the DDIM hot path cannot run "
                               "(diffusiondepth_amd has no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError(f"HipDenoiser needs a HIP device, got {self.device}")
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self.variant = variant
        h = ctypes.c_void_p()
        rc = self._lib.dd_create(ctypes.byref(h), idx, VARIANTS[variant])
        if rc != 0:
            raise RuntimeError(f"dd_create failed ({rc}): {self._lib.dd_last_error(None).decode()}")
        self._h = h
        self._have_schedule = False
        self._have_weights = False
        self._have_fpn = False
        self._have_neck = False
        self._cond_token = None      # (tensor, version, precision id) of the map the last condition() call returned
        # DDEPTH_STREAMS=S: dd_denoise / dd_denoise_backward run a batch of B >= 2 as S concurrent sub-batches on separate HIP streams
        # (dd_set_option "streams": +6..13 % throughput on MI355X, per-image results bit-identical).  The C library's default is 1; this
        # binding's is 2 (the whole GPU suite passes either way; DDEPTH_STREAMS=1 turns it off); bench.py pins 1 for its headline line,
        # whose roofline object is defined per launch on one stream.  (An apparent instability of the head forward under it was a host-side
        # pause in a short average: profiles/history/r02_run29_lanes_head_trace.md.)
        # hipGraph replay only in a process whose environment had the runtime's graph fast path switched off BEFORE this package was imported
        # (diffusiondepth_amd/__init__.py: GRAPH_REPLAY_SAFE; DDEPTH_GRAPH=0 / 1 overrides): the library's own default trusts the variable as it finds it
        # at dd_create, which this package may have exported too late to matter
        import diffusiondepth_amd as _pkg
        want_graph = os.environ.get("DDEPTH_GRAPH")
        self.graph_replay = bool(int(want_graph)) if want_graph not in (None, "") else bool(getattr(_pkg, "GRAPH_REPLAY_SAFE", False))
        self.set_option("graph", 1 if self.graph_replay else 0)
        self.n_streams = max(1, int(os.environ.get("DDEPTH_STREAMS", "2") or 2))
        if self.n_streams > 1:
            self.set_option("streams", self.n_streams)

    # -- plumbing -----------------------------------------------------------------------------
    def _ck(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): {self._lib.dd_last_error(self._h).decode()}")

    def close(self):
        if getattr(self, "_h", None):
            self._lib.dd_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def version(self) -> str:
        return self._lib.dd_version().decode()

    def set_option(self, key: str, value: int):
        self._ck(self._lib.dd_set_option(self._h, key.encode(), int(value)), f"dd_set_option({key})")

    def counter(self, key: str) -> int:
        v = ctypes.c_int64()
        self._ck(self._lib.dd_get_counter(self._h, key.encode(), ctypes.byref(v)), f"dd_get_counter({key})")
        return int(v.value)

    def last_loop_ms(self) -> float:
        ms = ctypes.c_float()
        self._ck(self._lib.dd_last_loop_ms(self._h, ctypes.byref(ms)), "dd_last_loop_ms")
        return float(ms.value)

    def layer_ms(self, layer: int):
        tot, cnt = ctypes.c_double(), ctypes.c_int64()
        self._ck(self._lib.dd_get_layer_ms(self._h, layer, ctypes.byref(tot), ctypes.byref(cnt)), "dd_get_layer_ms")
        return float(tot.value), int(cnt.value)

    # -- parameters ---------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, object], prefix: str = "", device_route: Optional[bool] = None):
        """sd: {key: tensor | ndarray} using the reference's key names below ``prefix`` (e.g.
        'depth_head.').  Keys the library does not own (backbone, convup_fp, num_batches_tracked) are ignored.  ``sd`` may hold any
        subset of the three parameter groups (model.* / depth_transform.* / conv_lateral.* + conv_up.*); each group must be complete.

        device_route: denoiser parameters (model.*) that are tensors on this device go to the library without leaving HBM
        (dd_set_weight_device: fp32 -> kernel layouts by pack kernels) instead of D2H copy + host-side packing -- what a training loop
        needs after every optimizer.step().  None = the environment variable DDEPTH_DEVICE_WEIGHTS (default "1": the route leaves the host
        packer's bytes in HBM, digest-equal -- host emulation and, since round 2, tests/test_zzz_gpu_device_weights.py on an MI355X)."""
        torch = _torch()
        if device_route is None:
            device_route = os.environ.get("DDEPTH_DEVICE_WEIGHTS", "1") == "1"
        n = 0
        owned = ("model.", "depth_transform.", "conv_lateral.", "conv_up.", "hahineck.")
        # of the neck only the executed convolutions travel (the attention / embedding parameters are dead in the reference: necks.py)
        neck_live = ("hahineck.lateral_convs.", "hahineck.conv_proj.", "hahineck.trans_proj.", "hahineck.conv_fusion.", "hahineck.trans_fusion.")
        saw_fpn = saw_model = False
        stream = ctypes.c_void_p(_stream_ptr(self.device))
        with torch.cuda.device(self.device):
            for k, v in sd.items():
                if not k.startswith(prefix):
                    continue
                name = k[len(prefix):]
                if not name.startswith(owned) or name.endswith("num_batches_tracked"):
                    continue
                if name.startswith("hahineck.") and not name.startswith(neck_live):
                    continue
                saw_fpn = saw_fpn or name.startswith("conv_") or name.startswith("hahineck.")
                self._have_neck = self._have_neck or name.startswith("hahineck.")
                saw_model = saw_model or name.startswith("model.")
                if device_route and name.startswith("model.") and isinstance(v, torch.Tensor) and v.device == self.device:
                    d = v.detach().to(torch.float32).contiguous()      # no-op for fp32 parameters; a temporary otherwise (same stream)
                    self._ck(self._lib.dd_set_weight_device(self._h, name.encode(), d.data_ptr(), d.numel(), stream),
                             f"dd_set_weight_device({name})")
                    n += 1
                    continue
                if isinstance(v, torch.Tensor):
                    v = v.detach().to("cpu", torch.float32).contiguous().numpy()
                a = np.ascontiguousarray(np.asarray(v, dtype=np.float32))
                self._ck(self._lib.dd_set_weight(self._h, name.encode(), a.ctypes.data_as(ctypes.c_void_p), a.size),
                         f"dd_set_weight({name})")
                n += 1
            self._ck(self._lib.dd_commit_weights(self._h, stream), "dd_commit_weights")
        self._have_weights = self._have_weights or saw_model
        self._have_fpn = self._have_fpn or saw_fpn
        if saw_fpn:
            self._cond_token = None       # new FPN weights: the map dd_condition left in the handle is stale (the other groups do not touch it)
        return n

    def weights_digest(self) -> int:
        """Test hook (dd_debug_weights_digest): digest of every packed denoiser weight buffer in HBM."""
        d = ctypes.c_uint64()
        with _torch().cuda.device(self.device):
            self._ck(self._lib.dd_debug_weights_digest(self._h, ctypes.byref(d)), "dd_debug_weights_digest")
        return int(d.value)

    def set_schedule(self, alphas_cumprod):
        torch = _torch()
        if isinstance(alphas_cumprod, torch.Tensor):
            alphas_cumprod = alphas_cumprod.detach().to("cpu", torch.float32).numpy()
        a = np.ascontiguousarray(np.asarray(alphas_cumprod, dtype=np.float32))
        self._ck(self._lib.dd_set_schedule(self._h, a.ctypes.data_as(ctypes.c_void_p), a.size), "dd_set_schedule")
        self._have_schedule = True

    # -- condition aggregation -----------------------------------------------------------------
    def condition(self, fp, precision="fp32", export=True, neck=False):
        """neck=True: the HAHI neck of the Swin-L heads first (dd_neck_condition: `fp` are the RAW backbone maps), then --
        The head's FPN (reference …res.py:108-118, …res_swin_add.py:117-127) on the 4 backbone maps ``fp`` (finest
        first; widths 64..512 for the Res variant, 192..1536 for Swin).  Returns the (B,256,h,w) fp32 condition map (None
        with export=False); the same map stays in the handle in kernel layout and is picked up -- without conversion --
        when the returned tensor is passed to denoise / denoise_once unchanged."""
        torch = _torch()
        if len(fp) != 4:
            raise ValueError("condition() expects the 4 pyramid levels of the ResNet backbone")
        fp = [_check_tensor(f, f"fp[{i}]", dtype=torch.float32) for i, f in enumerate(fp)]
        B = fp[0].shape[0]
        # pyramids dd_condition is built for: ResNet (res variant); Swin-L or MPViT-small (swin variant: same UpSample_add denoiser)
        allowed = [(192, 384, 768, 1536), (128, 216, 288, 288)] if self.variant == "swin" else [(64, 128, 256, 512)]
        widths = tuple(int(f.shape[1]) if f.dim() == 4 else -1 for f in fp)
        if widths not in allowed or any(f.dim() != 4 or f.shape[0] != B for f in fp):
            raise ValueError(f"fp must be 4 maps (B,C_i,h_i,w_i) with C in {allowed}, got {[tuple(f.shape) for f in fp]}")
        ptrs = (ctypes.c_void_p * 4)(*[f.data_ptr() for f in fp])
        hs = (ctypes.c_int * 4)(*[f.shape[2] for f in fp])
        ws = (ctypes.c_int * 4)(*[f.shape[3] for f in fp])
        out = torch.empty((B, 256, fp[0].shape[2], fp[0].shape[3]), device=self.device, dtype=torch.float32) if export else None
        pid = precision_id(precision)
        with torch.cuda.device(self.device):
            fn, what = (self._lib.dd_neck_condition, "dd_neck_condition") if neck else (self._lib.dd_condition, "dd_condition")
            self._ck(fn(self._h, ptrs, hs, ws, 4, B, out.data_ptr() if export else None, pid, _stream_ptr(self.device)), what)
        self._cond_token = (out, out._version, pid) if export else None
        return out

    def _cond_arg(self, cond, precision):
        """data pointer to hand to the library: None (= use the map dd_condition left in the handle) when ``cond`` is the
        unmodified tensor condition() returned for this precision."""
        tok = self._cond_token
        if tok is not None and tok[0] is cond and tok[1] == cond._version and tok[2] == precision_id(precision):
            return None
        self._cond_token = None           # an explicit map overwrites the handle's copy
        return cond.data_ptr()

    def suggested_batch(self, lat_h: int, lat_w: int) -> int:
        """Images per dd_denoise call that keep the chip full for a latent of lat_h x lat_w (throughput serving; the reference's own
        test() feeds one image at a time): a launch of the two large convolutions has ceil(h / 8) * ceil(w / 32) workgroups per image
        and the chip holds `resident_slots` of them (two per CU), so a lane needs about two rounds of them for the tail of one
        round to disappear under the next -- lanes x ceil(2 * slots / tiles per image).  KITTI 176x608: 6; NYU 114x152: 28
        (measured: KITTI B=4 -> 16 +5 %; NYU B=4 -> 16 +48 %, profiles/history/r03_*, r04_*)."""
        tiles = ((int(lat_h) + 7) // 8) * ((int(lat_w) + 31) // 32)
        slots = self.counter("resident_slots")
        return self.n_streams * max(1, -(-2 * slots // tiles))

    # -- hot path -----------------------------------------------------------------------------
    def denoise(self, x_T, cond, num_inference_steps: int, precision="fp32", out=None, keep_trajectory=False):
        """The T-step loop (one hipGraph).  ``keep_trajectory=True`` (training forward): the library keeps the state entering every step
        and ``self.last_trajectory_ticket`` names them; passed to ``denoise_backward`` it saves that call a second forward loop."""
        torch = _torch()
        x_T = _check_tensor(x_T, "x_T", dtype=torch.float32)
        cond_in = cond
        cond = _check_tensor(cond, "cond", dtype=torch.float32)
        if cond is not cond_in:
            self._cond_token = None
        B, C, h, w = x_T.shape
        if C != 16 or cond.dim() != 4 or cond.shape[0] != B or cond.shape[1] != 256:
            raise ValueError(f"x_T must be (B,16,h,w) and cond (B,256,ch,cw); got {tuple(x_T.shape)}, {tuple(cond.shape)}")
        out = torch.empty_like(x_T) if out is None else _check_tensor(out, "out", x_T.shape, torch.float32)
        self.last_trajectory_ticket = 0
        if keep_trajectory:
            self.set_option("keep_trajectory", 1)
        try:
            with torch.cuda.device(self.device):
                self._ck(self._lib.dd_denoise(self._h, x_T.data_ptr(), self._cond_arg(cond, precision), out.data_ptr(), B, h, w,
                                              cond.shape[2], cond.shape[3], int(num_inference_steps),
                                              precision_id(precision), _stream_ptr(self.device)), "dd_denoise")
            if keep_trajectory and precision != "naive_fp32":
                self.last_trajectory_ticket = self.counter("trajectory_ticket")
        finally:
            if keep_trajectory:
                self.set_option("keep_trajectory", 0)
        return out

    def denoise_trace(self, x_T, cond, num_inference_steps: int, precision="fp32"):
        """The loop of the reference's *Vis heads: (T,B,16,h,w) tensor of the sample after every step; the last entry is what
        denoise() returns."""
        torch = _torch()
        x_T = _check_tensor(x_T, "x_T", dtype=torch.float32)
        cond_in = cond
        cond = _check_tensor(cond, "cond", dtype=torch.float32)
        if cond is not cond_in:
            self._cond_token = None
        B, C, h, w = x_T.shape
        if C != 16 or cond.dim() != 4 or cond.shape[0] != B or cond.shape[1] != 256:
            raise ValueError(f"x_T must be (B,16,h,w) and cond (B,256,ch,cw); got {tuple(x_T.shape)}, {tuple(cond.shape)}")
        T = int(num_inference_steps)
        states = torch.empty((max(T, 1), B, 16, h, w), device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._ck(self._lib.dd_denoise_trace(self._h, x_T.data_ptr(), self._cond_arg(cond, precision), states.data_ptr(), B, h, w,
                                                cond.shape[2], cond.shape[3], T, precision_id(precision), _stream_ptr(self.device)),
                     "dd_denoise_trace")
        return states

    def denoise_once(self, x_t, t, cond, precision="fp32", keep_trajectory=False):
        """One epsilon-network evaluation.  ``keep_trajectory=True`` (training forward): ``self.last_trajectory_ticket`` names the
        activations this call leaves in the library; passed to ``denoise_once_backward`` it saves that call its forward recompute."""
        torch = _torch()
        x_t = _check_tensor(x_t, "x_t", dtype=torch.float32)
        cond = _check_tensor(cond, "cond", dtype=torch.float32)
        B, C, h, w = x_t.shape
        t = torch.as_tensor(t, device=self.device).to(torch.int64).reshape(-1)
        if t.numel() == 1 and B > 1:
            t = t.expand(B)
        t = _check_tensor(t, "t", (B,), torch.int64)
        out = torch.empty_like(x_t)
        self.last_trajectory_ticket = 0
        if keep_trajectory:
            self.set_option("keep_trajectory", 1)
        try:
            with torch.cuda.device(self.device):
                self._ck(self._lib.dd_denoise_once(self._h, x_t.data_ptr(), t.data_ptr(), self._cond_arg(cond, precision), out.data_ptr(),
                                                   B, h, w, cond.shape[2], cond.shape[3], precision_id(precision),
                                                   _stream_ptr(self.device)), "dd_denoise_once")
            if keep_trajectory and precision != "naive_fp32":
                self.last_trajectory_ticket = self.counter("trajectory_ticket")
        finally:
            if keep_trajectory:
                self.set_option("keep_trajectory", 0)
        return out

    # -- training: backward of one denoiser call ---------------------------------------------------
    PARAM_SHAPES = {
        "model.noise_embedding.0.weight": (64, 16, 3, 3), "model.noise_embedding.0.bias": (64,),
        "model.noise_embedding.1.weight": (64,), "model.noise_embedding.1.bias": (64,),
        "model.noise_embedding.3.weight": (256, 64, 3, 3), "model.noise_embedding.3.bias": (256,),
        "model.noise_embedding.4.weight": (256,), "model.noise_embedding.4.bias": (256,),
        "model.time_embedding.weight": (1280, 256),
        "model.pred.0.weight": (64, 256, 3, 3), "model.pred.0.bias": (64,),
        "model.pred.1.weight": (64,), "model.pred.1.bias": (64,),
        "model.pred.3.weight": (16, 64, 3, 3), "model.pred.3.bias": (16,),
        "model.pred.4.weight": (16,), "model.pred.4.bias": (16,),
    }

    SWIN_PARAM_SHAPES = {
        "model.upsample_fuse.convA.conv.weight": (256, 256, 3, 3), "model.upsample_fuse.convA.conv.bias": (256,),
        "model.upsample_fuse.convB.conv.weight": (256, 256, 3, 3), "model.upsample_fuse.convB.conv.bias": (256,),
    }

    def param_shapes(self):
        d = dict(self.PARAM_SHAPES)
        if self.variant == "swin":
            d.update(self.SWIN_PARAM_SHAPES)
        return d

    def denoise_once_backward(self, x_t, t, cond, grad_eps, precision="naive_fp32", need_grad_x=True, need_grad_cond=True,
                              trajectory_ticket=0):
        """VJP of one ScheduledCNNRefine.forward (what autograd computes for ``self.model(...)`` in the reference's
        training step): returns (grad_x, grad_cond); parameter gradients accumulate in the handle (``grads()``)."""
        torch = _torch()
        x_t = _check_tensor(x_t, "x_t", dtype=torch.float32)
        cond = _check_tensor(cond, "cond", dtype=torch.float32)
        grad_eps = _check_tensor(grad_eps, "grad_eps", x_t.shape, torch.float32)
        B, C, h, w = x_t.shape
        t = torch.as_tensor(t, device=self.device).to(torch.int64).reshape(-1)
        if t.numel() == 1 and B > 1:
            t = t.expand(B)
        t = _check_tensor(t, "t", (B,), torch.int64)
        gx = torch.empty_like(x_t) if need_grad_x else None
        gc = torch.empty_like(cond) if need_grad_cond else None
        if trajectory_ticket:
            self.set_option("use_trajectory", int(trajectory_ticket))      # the forward call's activations instead of a recompute
        with torch.cuda.device(self.device):
            self._ck(self._lib.dd_denoise_once_backward(
                self._h, x_t.data_ptr(), t.data_ptr(), cond.data_ptr(), grad_eps.data_ptr(),
                gx.data_ptr() if gx is not None else None, gc.data_ptr() if gc is not None else None,
                B, h, w, cond.shape[2], cond.shape[3], precision_id(precision), _stream_ptr(self.device)), "dd_denoise_once_backward")
        self._cond_token = None
        return gx, gc

    def denoise_backward(self, x_T, cond, grad_x0, num_inference_steps: int, precision="naive_fp32", need_grad_xT=False, need_grad_cond=True,
                         trajectory_ticket=0):
        """Backward of the whole DDIM loop (autograd through ``self.pipeline(...)`` in the reference's training step):
        returns (grad_xT, grad_cond); parameter gradients accumulate in the handle.  ``trajectory_ticket`` = the ticket of the forward
        ``denoise(..., keep_trajectory=True)`` call on the same inputs: the library then reads the per-step states that call kept (if they
        are still there and the parameters are unchanged) instead of running the forward loop again; 0 = always regenerate."""
        torch = _torch()
        x_T = _check_tensor(x_T, "x_T", dtype=torch.float32)
        cond = _check_tensor(cond, "cond", dtype=torch.float32)
        grad_x0 = _check_tensor(grad_x0, "grad_x0", x_T.shape, torch.float32)
        B, C, h, w = x_T.shape
        gx = torch.empty_like(x_T) if need_grad_xT else None
        gc = torch.empty_like(cond) if need_grad_cond else None
        if trajectory_ticket:
            self.set_option("use_trajectory", int(trajectory_ticket))
        with torch.cuda.device(self.device):
            self._ck(self._lib.dd_denoise_backward(
                self._h, x_T.data_ptr(), cond.data_ptr(), grad_x0.data_ptr(), gx.data_ptr() if gx is not None else None,
                gc.data_ptr() if gc is not None else None, B, h, w, cond.shape[2], cond.shape[3], int(num_inference_steps),
                precision_id(precision), _stream_ptr(self.device)), "dd_denoise_backward")
        self._cond_token = None
        return gx, gc

    def zero_grad(self):
        self._ck(self._lib.dd_zero_grad(self._h, _stream_ptr(self.device)), "dd_zero_grad")

    def grad(self, name: str):
        torch = _torch()
        out = torch.empty(self.param_shapes()[name], device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._ck(self._lib.dd_get_grad(self._h, name.encode(), out.data_ptr(), out.numel(), _stream_ptr(self.device)), f"dd_get_grad({name})")
        return out

    def grads(self):
        return {k: self.grad(k) for k in self.param_shapes()}

    def add_noise(self, x0, noise, t):
        torch = _torch()
        x0 = _check_tensor(x0, "x0", dtype=torch.float32)
        noise = _check_tensor(noise, "noise", x0.shape, torch.float32)
        B = x0.shape[0]
        t = _check_tensor(torch.as_tensor(t, device=self.device).to(torch.int64).reshape(-1), "t", (B,), torch.int64)
        out = torch.empty_like(x0)
        C, h, w = (x0.shape[1], x0.shape[2], x0.shape[3]) if x0.dim() == 4 else (1, 1, x0[0].numel())
        with torch.cuda.device(self.device):
            self._ck(self._lib.dd_add_noise(self._h, x0.data_ptr(), noise.data_ptr(), t.data_ptr(), out.data_ptr(),
                                            B, C, h, w, _stream_ptr(self.device)), "dd_add_noise")
        return out

    def encode(self, depth):
        torch = _torch()
        depth = _check_tensor(depth, "depth", dtype=torch.float32)
        B, C, H, W = depth.shape
        if C != 1:
            raise ValueError("depth must be (B,1,H,W)")
        out = torch.empty((B, 16, (H - 1) // 2 + 1, (W - 1) // 2 + 1), device=depth.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._ck(self._lib.dd_encode(self._h, depth.data_ptr(), out.data_ptr(), B, H, W, _stream_ptr(self.device)), "dd_encode")
        return out

    def decode(self, latent):
        torch = _torch()
        latent = _check_tensor(latent, "latent", dtype=torch.float32)
        B, C, h, w = latent.shape
        if C != 16:
            raise ValueError("latent must be (B,16,h,w)")
        out = torch.empty((B, 1, 2 * h, 2 * w), device=latent.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._ck(self._lib.dd_decode(self._h, latent.data_ptr(), out.data_ptr(), B, h, w, _stream_ptr(self.device)), "dd_decode")
        return out

    def debug_fetch(self, name: str, B: int, h: int, w: int):
        torch = _torch()
        C = {"y1": 64, "y2": 256, "y3": 64, "y4": 16}[name]
        out = torch.empty((B, C, h, w), device=self.device, dtype=torch.float32)
        self._ck(self._lib.dd_debug_fetch(self._h, name.encode(), out.data_ptr(), out.numel(), _stream_ptr(self.device)),
                 "dd_debug_fetch")
        return out
