"""TEST INFRASTRUCTURE -- CPU oracle (NumPy, fp64 arithmetic on fp32-computed sampling coordinates) of the reference's
NLSPN refinement stage and of the DCNv2 extension under it.  Not part of the product: only tests/,
__graft_entry__.smoke() and tools may import it.

Restates, function by function:
  sample()              mdmcn_im2col_bilinear                      src/model/deformconv/src/cuda/modulated_deform_im2col_cuda.cuh:23-54
  mdcn_columns()        modulated_deformable_im2col_gpu_kernel     ...cuh:126-194
  mdcn_forward()        modulated_deform_conv_cuda_forward         src/model/deformconv/src/cuda/modulated_deform_conv_cuda.cu:19-121
  mdcn_backward()       modulated_deform_conv_cuda_backward        ...cu:124-283 with the kernels ...cuh:196-328
  nlspn_offset_affinity()  NLSPN._get_offset_affinity              src/model/nlspnmodel.py:87-163
  nlspn_propagate()     NLSPN.forward's loop + _propagate_once     src/model/nlspnmodel.py:165-207

Pinned (tests/test_oracle_dcn.py) against the reference's own device code compiled for the host (oracle/dcn_ref.py ->
oracle/_ref/libref_dcn.so) on random cases, against the known answers of the reference's self-test
(src/model/deformconv/test.py: zero offset == nn.Conv2d, identity kernel, im2col_step invariance), and against goldens minted
by running the reference's NLSPN class on CPU (tests/golden/make_golden_nlspn.py).

Sampling coordinates are computed in fp32 exactly as the kernels do (int base converted to float, plus the fp32 offset) so that
floor() picks the same cell as the reference; everything downstream is fp64.
"""
from __future__ import annotations

import numpy as np

F64 = np.float64


def out_size(H, W, kh, kw, stride, pad, dil):
    return ((H + 2 * pad[0] - (dil[0] * (kh - 1) + 1)) // stride[0] + 1,
            (W + 2 * pad[1] - (dil[1] * (kw - 1) + 1)) // stride[1] + 1)


def _coords(offset_b, tap_i, tap_j, kw, Ho, Wo, stride, pad, dil):
    """fp32 sampling position of tap (i, j) for every output pixel of one (sample, deformable group): ...cuh:170-178."""
    k = tap_i * kw + tap_j
    hb = (np.arange(Ho, dtype=np.int64) * stride[0] - pad[0] + tap_i * dil[0]).astype(np.float32)[:, None]
    wb = (np.arange(Wo, dtype=np.int64) * stride[1] - pad[1] + tap_j * dil[1]).astype(np.float32)[None, :]
    return (hb + offset_b[2 * k].astype(np.float32)).astype(np.float32), (wb + offset_b[2 * k + 1].astype(np.float32)).astype(np.float32)


def _cell(hs, ws, H, W):
    """floor cell, the four bilinear weights and the per-corner validity masks of mdmcn_im2col_bilinear (...cuh:27-49) together
    with the kernel's range test (...cuh:179)."""
    inside = (hs > -1) & (ws > -1) & (hs < H) & (ws < W)
    hl = np.floor(hs).astype(np.int64)
    wl = np.floor(ws).astype(np.int64)
    lh = hs.astype(F64) - hl
    lw = ws.astype(F64) - wl
    hh, hw = 1.0 - lh, 1.0 - lw
    corners = []
    for (dy, dx, wt) in ((0, 0, hh * hw), (0, 1, hh * lw), (1, 0, lh * hw), (1, 1, lh * lw)):
        y, x = hl + dy, wl + dx
        ok = inside & (y >= 0) & (y <= H - 1) & (x >= 0) & (x <= W - 1)
        corners.append((np.clip(y, 0, H - 1), np.clip(x, 0, W - 1), wt, ok))
    return inside, hl, wl, lh, lw, corners


def sample(im, hs, ws):
    """Bilinear sample of one (H, W) plane at fp32 positions; zero outside (-1, H) x (-1, W)."""
    H, W = im.shape
    _, _, _, _, _, corners = _cell(hs, ws, H, W)
    v = np.zeros(hs.shape, F64)
    for y, x, wt, ok in corners:
        v += np.where(ok, wt * im[y, x].astype(F64), 0.0)
    return v


def mdcn_columns(inp, offset, mask, kh, kw, stride, pad, dil, dg):
    """col[c, i*kw+j, b, ho, wo] = mask * sample(inp[b, c], position of tap (i, j))."""
    B, C, H, W = inp.shape
    Ho, Wo = out_size(H, W, kh, kw, stride, pad, dil)
    K = kh * kw
    cpg = C // dg
    col = np.zeros((C, K, B, Ho, Wo), F64)
    for b in range(B):
        for g in range(dg):
            off = offset[b, g * 2 * K:(g + 1) * 2 * K]
            msk = mask[b, g * K:(g + 1) * K]
            for i in range(kh):
                for j in range(kw):
                    hs, ws = _coords(off, i, j, kw, Ho, Wo, stride, pad, dil)
                    for c in range(g * cpg, (g + 1) * cpg):
                        col[c, i * kw + j, b] = sample(inp[b, c], hs, ws) * msk[i * kw + j].astype(F64)
    return col


def mdcn_forward(inp, weight, bias, offset, mask, stride=(1, 1), pad=(0, 0), dil=(1, 1), group=1, dg=1, im2col_step=64):
    B, C, H, W = inp.shape
    Co, Ck, kh, kw = weight.shape
    step = min(B, im2col_step)
    if B % step:                                          # AT_ASSERTM, modulated_deform_conv_cuda.cu:58
        raise ValueError("batch must be a multiple of im2col_step")
    if C % group or Co % group or C != Ck * group:        # :60-72
        raise ValueError("channels / group mismatch")
    Ho, Wo = out_size(H, W, kh, kw, stride, pad, dil)
    col = mdcn_columns(inp, offset, mask, kh, kw, stride, pad, dil, dg).reshape(group, (C // group) * kh * kw, B, Ho, Wo)
    wg = weight.astype(F64).reshape(group, Co // group, Ck * kh * kw)
    out = np.einsum("gok,gkbhw->bgohw", wg, col).reshape(B, Co, Ho, Wo) + bias.astype(F64)[None, :, None, None]
    return out


def mdcn_backward(inp, weight, bias, offset, mask, grad_out, stride=(1, 1), pad=(0, 0), dil=(1, 1), group=1, dg=1,
                  pad_w_slip=True):
    """-> grad_input, grad_offset, grad_mask, grad_weight, grad_bias (fp64).
    pad_w_slip: the reference's col2im launcher hands pad_h to the kernel for BOTH paddings (...cuh:372), so grad_input is
    computed with pad_w := pad_h; True reproduces that (no effect when pad_h == pad_w, the only case NLSPN uses)."""
    B, C, H, W = inp.shape
    Co, Ck, kh, kw = weight.shape
    K = kh * kw
    Ho, Wo = out_size(H, W, kh, kw, stride, pad, dil)
    cpg = C // dg
    go = grad_out.astype(F64)
    wg = weight.astype(F64).reshape(group, Co // group, Ck * K)
    gog = go.reshape(B, group, Co // group, Ho, Wo)
    # columns = W^T . grad_out  (modulated_deform_conv_cuda.cu:218-223)
    gcol = np.einsum("gok,bgohw->gkbhw", wg, gog).reshape(C, K, B, Ho, Wo)
    g_in = np.zeros(inp.shape, F64)
    g_off = np.zeros(offset.shape, F64)
    g_msk = np.zeros(mask.shape, F64)
    col = np.zeros((C, K, B, Ho, Wo), F64)
    pad_im = (pad[0], pad[0]) if pad_w_slip else pad
    for b in range(B):
        for g in range(dg):
            off = offset[b, g * 2 * K:(g + 1) * 2 * K]
            msk = mask[b, g * K:(g + 1) * K].astype(F64)
            for i in range(kh):
                for j in range(kw):
                    k = i * kw + j
                    hs, ws = _coords(off, i, j, kw, Ho, Wo, stride, pad, dil)
                    inside, hl, wl, lh, lw, corners = _cell(hs, ws, H, W)
                    hs2, ws2 = _coords(off, i, j, kw, Ho, Wo, stride, pad_im, dil)
                    _, _, _, _, _, corners_im = _cell(hs2, ws2, H, W)
                    for c in range(g * cpg, (g + 1) * cpg):
                        im = inp[b, c].astype(F64)
                        v = [np.where(ok, im[y, x], 0.0) for (y, x, _, ok) in corners]
                        val = sum(np.where(ok, wt, 0.0) * vv for (_, _, wt, ok), vv in zip(corners, v))
                        col[c, k, b] = val * msk[k]
                        gc = gcol[c, k, b]
                        # offset / mask gradients: modulated_deformable_col2im_coord_gpu_kernel (...cuh:256-328) with
                        # mdmcn_get_coordinate_weight (...cuh:83-125): d/dh = (v3 - v1) * hw + (v4 - v2) * lw, d/dw = (v2 - v1) * hh + (v4 - v3) * lh
                        dh = np.where(inside, (v[2] - v[0]) * (1.0 - lw) + (v[3] - v[1]) * lw, 0.0)
                        dw = np.where(inside, (v[1] - v[0]) * (1.0 - lh) + (v[3] - v[2]) * lh, 0.0)
                        g_off[b, g * 2 * K + 2 * k] += dh * gc * msk[k]
                        g_off[b, g * 2 * K + 2 * k + 1] += dw * gc * msk[k]
                        g_msk[b, g * K + k] += np.where(inside, gc * val, 0.0)
                        # input gradient: modulated_deformable_col2im_gpu_kernel (...cuh:196-254): scatter of the bilinear weights
                        top = gc * msk[k]
                        for (y, x, wt, ok) in corners_im:
                            np.add.at(g_in[b, c], (y[ok], x[ok]), (wt * top)[ok])
    colg = col.reshape(group, (C // group) * K, B, Ho, Wo)
    g_w = np.einsum("bgohw,gkbhw->gok", gog, colg).reshape(weight.shape)                    # :268-272
    g_b = go.sum(axis=(0, 2, 3))                                                            # :273
    return g_in, g_off, g_msk, g_w, g_b


# ------------------------------------------------------------------------------------------------------------------------
# NLSPN (src/model/nlspnmodel.py)

def nlspn_offset_affinity(offset_aff, confidence, aff_scale_const, affinity="TGASS", k_f=3, conf_prop=True, legacy=False,
                          w_conf=1.0, b=0.0):
    """NLSPN._get_offset_affinity after ``offset_aff = self.conv_offset_aff(guidance)`` (nlspnmodel.py:90-163).
    offset_aff (B, 3*num, H, W), confidence (B, 1, H, W) or None -> offset (B, 2*(num+1), H, W), aff (B, num+1, H, W)."""
    B, C3, H, W = offset_aff.shape
    num = k_f * k_f - 1
    assert C3 == 3 * num
    idx_ref = num // 2
    oa = offset_aff.astype(F64)
    # torch.cat((o1, o2), 1).view(B, num, 2, H, W): neighbour n gets channels (2n, 2n+1) of the first 2*num channels (:92-95)
    off = oa[:, :2 * num].reshape(B, num, 2, H, W)
    off = np.concatenate([off[:, :idx_ref], np.zeros((B, 1, 2, H, W), F64), off[:, idx_ref:]], axis=1)       # (:96-99)
    aff = oa[:, 2 * num:].copy()
    if affinity in ("AS", "ASS"):
        pass
    elif affinity == "TC":
        aff = np.tanh(aff) / float(aff_scale_const)                                        # (:104)
    elif affinity == "TGASS":
        aff = np.tanh(aff) / (float(aff_scale_const) + 1e-8)                               # (:106)
    else:
        raise NotImplementedError(affinity)
    if conf_prop:
        conf = []
        for idx in range(num + 1):
            ww, hh = idx % k_f, idx // k_f
            if ww == (k_f - 1) / 2 and hh == (k_f - 1) / 2:
                continue
            if legacy:                                 # in place on a detach()ed VIEW of `offset`: the returned offsets move too (:126-134)
                off[:, idx, 0] += hh - (k_f - 1) / 2
                off[:, idx, 1] += ww - (k_f - 1) / 2
            o2 = off[:, idx].astype(np.float32)        # (B, 2, H, W)
            # 1x1 modulated deformable conv of the confidence, padding 0, mask of ones, weight w_conf, bias b (:136-141)
            c = mdcn_forward(confidence.astype(np.float32), np.full((1, 1, 1, 1), w_conf, np.float32), np.full((1,), b, np.float32),
                             o2, np.ones((B, 1, H, W), np.float32))
            conf.append(c)
        aff = aff * np.concatenate(conf, axis=1)                                           # (:143-144)
    s = np.abs(aff).sum(axis=1, keepdims=True) + 1e-4                                      # (:147-148)
    if affinity in ("ASS", "TGASS"):
        s = np.where(s < 1.0, 1.0, s)                                                      # (:150-151)
    if affinity in ("AS", "ASS", "TGASS"):
        aff = aff / s                                                                      # (:153-154)
    ref = 1.0 - aff.sum(axis=1, keepdims=True)                                             # (:156-157)
    aff = np.concatenate([aff[:, :idx_ref], ref, aff[:, idx_ref:]], axis=1)                # (:159-161)
    return off.reshape(B, 2 * (num + 1), H, W), aff


def nlspn_propagate(feat_init, offset, aff, feat_fix=None, prop_time=18, preserve_input=False, k_f=3, w=None, b=0.0):
    """The propagation loop of NLSPN.forward (nlspnmodel.py:186-205): prop_time modulated deformable 3x3 convolutions of the
    one-channel depth with the affinities as modulation.  -> (feat_result, [feat after every iteration])."""
    B, _, H, W = feat_init.shape
    w = np.ones((1, 1, k_f, k_f), np.float32) if w is None else np.asarray(w, np.float32).reshape(1, 1, k_f, k_f)
    bias = np.full((1,), b, np.float32)
    pad = (k_f - 1) // 2
    feat = feat_init.astype(F64)
    if preserve_input:
        m = ((feat_fix > 0).sum(axis=1, keepdims=True) > 0).astype(F64)                    # (:189-191)
    feats = []
    off32, aff32 = offset.astype(np.float32), aff.astype(np.float32)
    for _ in range(prop_time):
        if preserve_input:
            feat = (1.0 - m) * feat + m * feat_fix.astype(F64)                             # (:199-201)
        feat = mdcn_forward(feat, w, bias, off32, aff32, pad=(pad, pad))                   # _propagate_once (:165-171)
        feats.append(feat)
    return feat, feats
