"""Drop-in for the reference's ``DCN`` extension module and its autograd wrapper, over the C ABI of include/ddepth_dcn.h.

Reference interfaces mirrored here (same names, argument order and error behaviour):
  DCN.modulated_deform_conv_forward / _backward   src/model/deformconv/src/vision.cpp:10-11 (pybind11; at::Tensor in / out)
  ModulatedDeformConvFunction                     src/model/modulated_deform_conv_func.py:15-56
so a reference checkout switches over with ``import diffusiondepth_amd.dcn as DCN``.  All arithmetic runs in
libddepth_hip.so (csrc/dd_dcn.hip); like the reference's extension there is no CPU implementation
(src/model/deformconv/src/modulated_deform_conv.h:39-43 raises "Not implemented on the CPU").
"""
from __future__ import annotations

import ctypes

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from . import backend

# every symbol include/ddepth_dcn.h declares (checked by tests/test_abi.py)
ABI_SYMBOLS = ["dd_dcn_last_error", "dd_dcn_forward", "dd_dcn_backward", "dd_nlspn_offset_affinity", "dd_nlspn_guided_offset_affinity",
               "dd_nlspn_workspace_bytes",
               "dd_nlspn_propagate"]

_bound = None


def _lib():
    global _bound
    if _bound is None:
        lib = backend.load_library()
        c_int, c_vp = ctypes.c_int, ctypes.c_void_p
        lib.dd_dcn_last_error.restype, lib.dd_dcn_last_error.argtypes = ctypes.c_char_p, []
        lib.dd_dcn_forward.restype, lib.dd_dcn_forward.argtypes = c_int, [c_vp] * 6 + [c_int] * 16 + [c_vp]
        lib.dd_dcn_backward.restype, lib.dd_dcn_backward.argtypes = c_int, [c_vp] * 11 + [c_int] * 16 + [c_vp]
        lib.dd_nlspn_offset_affinity.restype, lib.dd_nlspn_offset_affinity.argtypes = c_int, [c_vp] * 7 + [c_int] * 7 + [c_vp]
        lib.dd_nlspn_guided_offset_affinity.restype, lib.dd_nlspn_guided_offset_affinity.argtypes = c_int, [c_vp] * 9 + [c_int] * 9 + [c_vp]
        lib.dd_nlspn_workspace_bytes.restype = c_int
        lib.dd_nlspn_workspace_bytes.argtypes = [c_int, c_int, c_int, c_int, ctypes.POINTER(ctypes.c_int64)]
        lib.dd_nlspn_propagate.restype, lib.dd_nlspn_propagate.argtypes = c_int, [c_vp] * 8 + [c_int] * 6 + [c_vp]
        _bound = lib
    return _bound


def _ck(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {_lib().dd_dcn_last_error().decode()}")


def _dev_f32(t, name):
    """The reference asserts instead of fixing (AT_ASSERTM(input.is_contiguous()), AT_ASSERTM(input.type().is_cuda()),
    modulated_deform_conv_cuda.cu:39-46)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a CUDA tensor (Not implemented on the CPU)")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name} must be float32, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} tensor has to be contiguous")
    return t


def require_hip(t, what):
    """The module-level guard of nlspn.NLSPN.forward: like the reference's extension, there is no CPU path."""
    if not t.is_cuda:
        raise RuntimeError(f"{what} runs only on a HIP device (the reference's DCN extension has no CPU path either: "
                           "src/model/deformconv/src/modulated_deform_conv.h:39-43)")


def _stream(t):
    return ctypes.c_void_p(int(torch.cuda.current_stream(t.device).cuda_stream))


def _check_shapes(input, weight, bias, offset, mask, kernel_h, kernel_w, stride, pad, dil, group, dg):
    if input.dim() != 4 or weight.dim() != 4:
        raise RuntimeError("input and weight must be 4-D")
    B, C, H, W = input.shape
    Co, Ck, kh, kw = weight.shape
    if (kh, kw) != (kernel_h, kernel_w):                       # modulated_deform_conv_cuda.cu:67-68
        raise RuntimeError(f"Input shape and kernel shape wont match: ({kh} x {kw} vs {kernel_h} x {kernel_w}).")
    if C != Ck * group:                                        # :70-71
        raise RuntimeError(f"Input shape and kernel channels wont match: ({C} vs {Ck * group}).")
    Ho = (H + 2 * pad[0] - (dil[0] * (kh - 1) + 1)) // stride[0] + 1
    Wo = (W + 2 * pad[1] - (dil[1] * (kw - 1) + 1)) // stride[1] + 1
    if tuple(offset.shape) != (B, dg * 2 * kh * kw, Ho, Wo):
        raise RuntimeError(f"offset must be {(B, dg * 2 * kh * kw, Ho, Wo)}, got {tuple(offset.shape)}")
    if tuple(mask.shape) != (B, dg * kh * kw, Ho, Wo):
        raise RuntimeError(f"mask must be {(B, dg * kh * kw, Ho, Wo)}, got {tuple(mask.shape)}")
    if bias.numel() != Co:
        raise RuntimeError(f"bias must have {Co} elements")
    return B, C, H, W, Co, Ho, Wo


def modulated_deform_conv_forward(input, weight, bias, offset, mask, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                                  dilation_h, dilation_w, group, deformable_group, im2col_step):
    """DCN.modulated_deform_conv_forward (vision.cpp:10)."""
    input, weight, bias, offset, mask = (_dev_f32(t, n) for t, n in ((input, "input"), (weight, "weight"), (bias, "bias"),
                                                                      (offset, "offset"), (mask, "mask")))
    B, C, H, W, Co, Ho, Wo = _check_shapes(input, weight, bias, offset, mask, kernel_h, kernel_w, (stride_h, stride_w), (pad_h, pad_w),
                                           (dilation_h, dilation_w), group, deformable_group)
    out = torch.empty((B, Co, Ho, Wo), device=input.device, dtype=torch.float32)
    with torch.cuda.device(input.device):
        _ck(_lib().dd_dcn_forward(input.data_ptr(), weight.data_ptr(), bias.data_ptr(), offset.data_ptr(), mask.data_ptr(), out.data_ptr(),
                                  B, C, H, W, Co, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group,
                                  deformable_group, im2col_step, _stream(input)), "dd_dcn_forward")
    return out


def modulated_deform_conv_backward(input, weight, bias, offset, mask, grad_output, kernel_h, kernel_w, stride_h, stride_w, pad_h, pad_w,
                                   dilation_h, dilation_w, group, deformable_group, im2col_step, needs=(True,) * 5):
    """DCN.modulated_deform_conv_backward (vision.cpp:11) -> [grad_input, grad_offset, grad_mask, grad_weight, grad_bias].
    ``needs`` (not in the reference) skips outputs nobody asked for; skipped entries are None."""
    input, weight, bias, offset, mask = (_dev_f32(t, n) for t, n in ((input, "input"), (weight, "weight"), (bias, "bias"),
                                                                      (offset, "offset"), (mask, "mask")))
    grad_output = _dev_f32(grad_output.contiguous(), "grad_output")
    B, C, H, W, Co, Ho, Wo = _check_shapes(input, weight, bias, offset, mask, kernel_h, kernel_w, (stride_h, stride_w), (pad_h, pad_w),
                                           (dilation_h, dilation_w), group, deformable_group)
    if tuple(grad_output.shape) != (B, Co, Ho, Wo):            # modulated_deform_conv_cuda.cu:187-194
        raise RuntimeError(f"Input shape and grad_out shape wont match: {(B, Co, Ho, Wo)} vs {tuple(grad_output.shape)}")
    outs = [torch.empty_like(t) if n else None for t, n in zip((input, offset, mask, weight, bias), needs)]
    ptr = [o.data_ptr() if o is not None else None for o in outs]
    with torch.cuda.device(input.device):
        _ck(_lib().dd_dcn_backward(input.data_ptr(), weight.data_ptr(), bias.data_ptr(), offset.data_ptr(), mask.data_ptr(),
                                   grad_output.data_ptr(), ptr[0], ptr[1], ptr[2], ptr[3], ptr[4], B, C, H, W, Co, kernel_h, kernel_w,
                                   stride_h, stride_w, pad_h, pad_w, dilation_h, dilation_w, group, deformable_group, im2col_step,
                                   _stream(input)), "dd_dcn_backward")
    return outs


class ModulatedDeformConvFunction(Function):
    """src/model/modulated_deform_conv_func.py:15-56, same apply() signature."""

    @staticmethod
    def forward(ctx, input, offset, mask, weight, bias, stride, padding, dilation, groups, deformable_groups, im2col_step):
        ctx.stride = _pair(stride)
        ctx.padding = _pair(padding)
        ctx.dilation = _pair(dilation)
        ctx.kernel_size = _pair(weight.shape[2:4])
        ctx.groups = groups
        ctx.deformable_groups = deformable_groups
        ctx.im2col_step = im2col_step
        output = modulated_deform_conv_forward(input, weight, bias, offset, mask, ctx.kernel_size[0], ctx.kernel_size[1],
                                               ctx.stride[0], ctx.stride[1], ctx.padding[0], ctx.padding[1], ctx.dilation[0],
                                               ctx.dilation[1], ctx.groups, ctx.deformable_groups, ctx.im2col_step)
        ctx.save_for_backward(input, offset, mask, weight, bias)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, mask, weight, bias = ctx.saved_tensors
        needs = ctx.needs_input_grad[:5]        # (input, offset, mask, weight, bias)
        gi, go, gm, gw, gb = modulated_deform_conv_backward(
            input, weight, bias, offset, mask, grad_output, ctx.kernel_size[0], ctx.kernel_size[1], ctx.stride[0], ctx.stride[1],
            ctx.padding[0], ctx.padding[1], ctx.dilation[0], ctx.dilation[1], ctx.groups, ctx.deformable_groups, ctx.im2col_step,
            needs=(needs[0], needs[1], needs[2], needs[3], needs[4]))
        return gi, go, gm, gw, gb, None, None, None, None, None, None


AFFINITY_ID = {"AS": 0, "ASS": 1, "TC": 2, "TGASS": 3}


def nlspn_offset_affinity(offset_aff, confidence, aff_scale_const, w_conf, b_conf, k_f, affinity, conf_prop, legacy):
    """dd_nlspn_offset_affinity: NLSPN._get_offset_affinity after its convolution (src/model/nlspnmodel.py:90-163) in one kernel."""
    offset_aff = _dev_f32(offset_aff.contiguous(), "offset_aff")
    B, C3, H, W = offset_aff.shape
    num = k_f * k_f - 1
    if C3 != 3 * num:
        raise RuntimeError(f"offset_aff must have {3 * num} channels, got {C3}")
    if conf_prop:
        if confidence is None:
            raise AssertionError("conf_prop needs a confidence map")          # nlspnmodel.py:180
        confidence = _dev_f32(confidence.contiguous(), "confidence")
        if tuple(confidence.shape) != (B, 1, H, W):
            raise RuntimeError(f"confidence must be {(B, 1, H, W)}, got {tuple(confidence.shape)}")
    offset = torch.empty((B, 2 * (num + 1), H, W), device=offset_aff.device, dtype=torch.float32)
    aff = torch.empty((B, num + 1, H, W), device=offset_aff.device, dtype=torch.float32)
    dp = lambda t: _dev_f32(t.detach().contiguous(), "parameter").data_ptr()
    with torch.cuda.device(offset_aff.device):
        _ck(_lib().dd_nlspn_offset_affinity(offset_aff.data_ptr(), confidence.data_ptr() if conf_prop else None, dp(aff_scale_const),
                                            dp(w_conf), dp(b_conf), offset.data_ptr(), aff.data_ptr(), B, H, W, k_f, AFFINITY_ID[affinity],
                                            int(bool(conf_prop)), int(bool(legacy)), _stream(offset_aff)), "dd_nlspn_offset_affinity")
    return offset, aff


def guided_supported(ch_g, k_g, k_f):
    """Geometry dd_nlspn_guided_offset_affinity is built for (NLSPNModel's: src/model/nlspnmodel.py:215,287-288)."""
    return ch_g == 8 and k_g == 3 and k_f == 3


def nlspn_guided_offset_affinity(guidance, conv_weight, conv_bias, confidence, aff_scale_const, w_conf, b_conf, k_g, k_f, affinity,
                                 conf_prop, legacy):
    """dd_nlspn_guided_offset_affinity: conv_offset_aff + the whole of _get_offset_affinity (src/model/nlspnmodel.py:87-163) in one kernel."""
    guidance = _dev_f32(guidance.contiguous(), "guidance")
    B, ch_g, H, W = guidance.shape
    num = k_f * k_f - 1
    if tuple(conv_weight.shape) != (3 * num, ch_g, k_g, k_g):
        raise RuntimeError(f"conv_offset_aff.weight must be {(3 * num, ch_g, k_g, k_g)}, got {tuple(conv_weight.shape)}")
    if conf_prop:
        if confidence is None:
            raise AssertionError("conf_prop needs a confidence map")          # nlspnmodel.py:180
        confidence = _dev_f32(confidence.contiguous(), "confidence")
        if tuple(confidence.shape) != (B, 1, H, W):
            raise RuntimeError(f"confidence must be {(B, 1, H, W)}, got {tuple(confidence.shape)}")
    offset = torch.empty((B, 2 * (num + 1), H, W), device=guidance.device, dtype=torch.float32)
    aff = torch.empty((B, num + 1, H, W), device=guidance.device, dtype=torch.float32)
    dp = lambda t: _dev_f32(t.detach().contiguous(), "parameter").data_ptr()
    with torch.cuda.device(guidance.device):
        _ck(_lib().dd_nlspn_guided_offset_affinity(guidance.data_ptr(), dp(conv_weight), dp(conv_bias),
                                                   confidence.data_ptr() if conf_prop else None, dp(aff_scale_const), dp(w_conf), dp(b_conf),
                                                   offset.data_ptr(), aff.data_ptr(), B, ch_g, H, W, k_g, k_f, AFFINITY_ID[affinity],
                                                   int(bool(conf_prop)), int(bool(legacy)), _stream(guidance)), "dd_nlspn_guided_offset_affinity")
    return offset, aff


def nlspn_propagate(feat_init, offset, aff, feat_fix, w, b, k_f, prop_time, preserve_input):
    """dd_nlspn_propagate: the propagation loop of NLSPN.forward (src/model/nlspnmodel.py:186-205).
    -> (prop_time, B, 1, H, W) tensor holding the result of every iteration."""
    feat_init, offset, aff = _dev_f32(feat_init.contiguous(), "feat_init"), _dev_f32(offset.contiguous(), "offset"), _dev_f32(aff.contiguous(), "aff")
    B, C, H, W = feat_init.shape
    K = k_f * k_f
    if C != 1:
        raise AssertionError(f"only tested with ch_f == 1 but {C}")           # nlspnmodel.py:30
    if tuple(offset.shape) != (B, 2 * K, H, W) or tuple(aff.shape) != (B, K, H, W):
        raise RuntimeError(f"offset / aff must be {(B, 2 * K, H, W)} / {(B, K, H, W)}, got {tuple(offset.shape)} / {tuple(aff.shape)}")
    ws = None
    if preserve_input:
        feat_fix = _dev_f32(feat_fix.contiguous(), "feat_fix")
        assert feat_init.shape == feat_fix.shape                              # nlspnmodel.py:188
        nbytes = ctypes.c_int64()
        _ck(_lib().dd_nlspn_workspace_bytes(B, H, W, 1, ctypes.byref(nbytes)), "dd_nlspn_workspace_bytes")
        ws = torch.empty(nbytes.value // 4, device=feat_init.device, dtype=torch.float32)
    feats = torch.empty((prop_time, B, 1, H, W), device=feat_init.device, dtype=torch.float32)
    dp = lambda t: _dev_f32(t.detach().contiguous(), "parameter").data_ptr()
    with torch.cuda.device(feat_init.device):
        _ck(_lib().dd_nlspn_propagate(feat_init.data_ptr(), offset.data_ptr(), aff.data_ptr(), feat_fix.data_ptr() if preserve_input else None,
                                      dp(w), dp(b), feats.data_ptr(), ws.data_ptr() if ws is not None else None, B, H, W, k_f,
                                      int(prop_time), int(bool(preserve_input)), _stream(feat_init)), "dd_nlspn_propagate")
    return feats
