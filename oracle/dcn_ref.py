"""TEST INFRASTRUCTURE: the reference's DCNv2 extension run on the host.

``oracle/_ref/libref_dcn.so`` holds the reference's own device code (modulated_deform_im2col_cuda.cuh, compiled for the CPU by
oracle/ref_dcn/build_ref.py).  This module adds the host orchestration around the three kernels exactly as the reference's
``modulated_deform_conv_cuda_forward`` / ``_backward`` do it (src/model/deformconv/src/cuda/modulated_deform_conv_cuda.cu:19-121,
:124-283): per im2col_step chunk, im2col -> per-group addmm; backward: columns = W^T . grad_out -> col2im_coord, col2im, im2col ->
grad_weight / grad_bias GEMMs.  The matrix products are NumPy fp32 (``at::addmm`` / ``at::mm`` in the reference).

Used by: tests/test_oracle_dcn.py (pins oracle/dcn_oracle.py to the reference's arithmetic) and
tests/golden/make_golden_nlspn.py (stands in for the CUDA-only ``DCN`` module when the reference's NLSPN class is run on CPU).
Never imported by the product package.
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def library_path() -> str:
    return os.path.join(_HERE, "_ref", "libref_dcn.so")


def available() -> bool:
    if os.path.exists(library_path()):
        return True
    from oracle.ref_dcn import build_ref
    return build_ref.reference_available()


def _lib():
    global _LIB
    if _LIB is None:
        from oracle.ref_dcn import build_ref
        path = build_ref.build(verbose=False) or library_path()
        if not os.path.exists(path):
            raise RuntimeError("oracle/_ref/libref_dcn.so missing and /root/reference absent: run oracle/ref_dcn/build_ref.py in the build container")
        _LIB = ctypes.CDLL(path)
        _LIB.ref_dcn_origin.restype = ctypes.c_char_p
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def out_size(H, W, kh, kw, stride, pad, dil):
    Ho = (H + 2 * pad[0] - (dil[0] * (kh - 1) + 1)) // stride[0] + 1      # modulated_deform_conv_cuda.cu:75-76
    Wo = (W + 2 * pad[1] - (dil[1] * (kw - 1) + 1)) // stride[1] + 1
    return Ho, Wo


def im2col(inp, offset, mask, kh, kw, stride, pad, dil, dg):
    inp, offset, mask = _f32(inp), _f32(offset), _f32(mask)
    B, C, H, W = inp.shape
    Ho, Wo = out_size(H, W, kh, kw, stride, pad, dil)
    col = np.empty((C * kh * kw, B * Ho * Wo), np.float32)
    _lib().ref_mdcn_im2col(_p(inp), _p(offset), _p(mask), B, C, H, W, Ho, Wo, kh, kw, pad[0], pad[1], stride[0], stride[1],
                           dil[0], dil[1], dg, _p(col))
    return col


def forward(inp, weight, bias, offset, mask, stride=(1, 1), pad=(0, 0), dil=(1, 1), group=1, dg=1, im2col_step=64):
    """modulated_deform_conv_cuda_forward (modulated_deform_conv_cuda.cu:19-121)."""
    inp, weight, bias, offset, mask = map(_f32, (inp, weight, bias, offset, mask))
    B, C, H, W = inp.shape
    Co, Ck, kh, kw = weight.shape
    step = min(B, im2col_step)
    assert B % step == 0 and C % group == 0 and Co % group == 0 and C == Ck * group
    Ho, Wo = out_size(H, W, kh, kw, stride, pad, dil)
    out = np.empty((B * Ho * Wo, Co), np.float32)
    wg = weight.reshape(group, Co // group, Ck * kh * kw)
    bg = bias.reshape(group, Co // group)
    for n in range(B // step):
        sl = slice(n * step, (n + 1) * step)
        col = im2col(inp[sl], offset[sl], mask[sl], kh, kw, stride, pad, dil, dg)          # (C*K, step*Ho*Wo)
        colg = col.reshape(group, (C // group) * kh * kw, step * Ho * Wo)
        o = out[n * step * Ho * Wo:(n + 1) * step * Ho * Wo].reshape(step * Ho * Wo, group, Co // group)
        for g in range(group):
            o[:, g, :] = bg[g][None, :] + colg[g].T @ wg[g].T                               # at::addmm (:113)
    return np.ascontiguousarray(out.reshape(B, Ho, Wo, Co).transpose(0, 3, 1, 2))         # (:118)


def backward(inp, weight, bias, offset, mask, grad_out, stride=(1, 1), pad=(0, 0), dil=(1, 1), group=1, dg=1, im2col_step=64):
    """modulated_deform_conv_cuda_backward (modulated_deform_conv_cuda.cu:124-283) ->
    (grad_input, grad_offset, grad_mask, grad_weight, grad_bias)."""
    inp, weight, bias, offset, mask, grad_out = map(_f32, (inp, weight, bias, offset, mask, grad_out))
    B, C, H, W = inp.shape
    Co, Ck, kh, kw = weight.shape
    step = min(B, im2col_step)
    assert B % step == 0
    Ho, Wo = out_size(H, W, kh, kw, stride, pad, dil)
    assert grad_out.shape == (B, Co, Ho, Wo)
    gi, gw, gb = np.zeros_like(inp), np.zeros_like(weight), np.zeros_like(bias)
    go_, gm_ = np.zeros_like(offset), np.zeros_like(mask)
    wg = weight.reshape(group, Co // group, Ck * kh * kw)
    gwg = gw.reshape(group, Co // group, Ck * kh * kw)
    gbg = gb.reshape(group, Co // group)
    lib = _lib()
    for n in range(B // step):
        sl = slice(n * step, (n + 1) * step)
        gog = grad_out[sl].reshape(step, group, Co // group, Ho, Wo)
        col = np.empty((C * kh * kw, step * Ho * Wo), np.float32)
        colg = col.reshape(group, (C // group) * kh * kw, step * Ho * Wo)
        gom = [np.ascontiguousarray(gog[:, g].transpose(1, 0, 2, 3)).reshape(Co // group, step * Ho * Wo) for g in range(group)]
        for g in range(group):
            colg[g] = wg[g].T @ gom[g]                                                      # (:222)
        i_n, o_n, m_n = _f32(inp[sl]), _f32(offset[sl]), _f32(mask[sl])
        go_n, gm_n, gi_n = np.zeros_like(o_n), np.zeros_like(m_n), np.zeros_like(i_n)
        a = (step, C, H, W, Ho, Wo, kh, kw, pad[0], pad[1], stride[0], stride[1], dil[0], dil[1], dg)
        lib.ref_mdcn_col2im_coord(_p(col), _p(i_n), _p(o_n), _p(m_n), *a, _p(go_n), _p(gm_n))     # (:226-236)
        lib.ref_mdcn_col2im(_p(col), _p(o_n), _p(m_n), *a, _p(gi_n))                                # (:238-246)
        go_[sl], gm_[sl], gi[sl] = go_n, gm_n, gi_n
        col2 = im2col(i_n, o_n, m_n, kh, kw, stride, pad, dil, dg)                                   # (:249-257)
        col2g = col2.reshape(group, (C // group) * kh * kw, step * Ho * Wo)
        for g in range(group):
            gwg[g] += gom[g] @ col2g[g].T                                                    # at::addmm (:272)
            gbg[g] += gom[g] @ np.ones(step * Ho * Wo, np.float32)                           # at::addmv (:273)
    return gi, go_, gm_, gw, gb
