#!/usr/bin/env python3
"""Mint known-answer vectors for the NLSPN refinement stage from the REFERENCE's own code.

Build container only (needs /root/reference).  What runs:
  * the reference's ``NLSPN`` module and ``ModulatedDeformConvFunction`` -- imported verbatim from
    /root/reference/src/model/nlspnmodel.py and modulated_deform_conv_func.py;
  * under them, instead of the CUDA-only ``DCN`` extension module, a stand-in with the same two functions
    (src/model/deformconv/src/vision.cpp:10-11) that runs the reference's own DCNv2 device code compiled for the host
    (oracle/dcn_ref.py -> oracle/_ref/libref_dcn.so, built by oracle/ref_dcn/build_ref.py from the reference header where it lies).
So every number stored here was produced by reference arithmetic; nothing of this repository's oracle or kernels is involved.

The reference zero-initialises conv_offset_aff (nlspnmodel.py:55-56: all offsets 0, all affinities 0), which would pin nothing, so
the cases draw seeded weights.  Inputs ARE stored (the cases are small).  Re-run:  python tests/golden/make_golden_nlspn.py

Cases -> tests/golden/nlspn_<name>.npz
  tgass      defaults of src/config.py:78-112 (TGASS, conf_prop, prop_time 18, 3x3), B=2, 20x28, forward + autograd gradients
  preserve   --preserve_input (sparse feat_fix), affinity ASS, prop_time 6
  as_noconf  affinity AS, --no_conf, prop_time 4
  tc_legacy  affinity TC with --legacy (offsets shifted in place), prop_time 3
  k5         prop_kernel 5 (24 neighbours), TGASS, prop_time 2
  dcn_*      plain modulated_deform_conv forward/backward cases (groups, deformable groups, stride, dilation) straight from the
             reference kernels, for the general DCNv2 entry points
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

from ref_import import REF_SRC, _install_stubs  # noqa: E402
from oracle import dcn_ref  # noqa: E402


def _install_dcn_standin():
    """``import DCN`` (modulated_deform_conv_func.py:13) -> the reference kernels on the host."""
    def fwd(input, weight, bias, offset, mask, kh, kw, sh, sw, ph, pw, dh, dw, group, dg, step):
        y = dcn_ref.forward(input.detach().numpy(), weight.detach().numpy(), bias.detach().numpy(), offset.detach().numpy(),
                            mask.detach().numpy(), (sh, sw), (ph, pw), (dh, dw), group, dg, step)
        return torch.from_numpy(y)

    def bwd(input, weight, bias, offset, mask, grad_output, kh, kw, sh, sw, ph, pw, dh, dw, group, dg, step):
        g = dcn_ref.backward(input.detach().numpy(), weight.detach().numpy(), bias.detach().numpy(), offset.detach().numpy(),
                             mask.detach().numpy(), grad_output.contiguous().numpy(), (sh, sw), (ph, pw), (dh, dw), group, dg, step)
        return [torch.from_numpy(a) for a in g]

    m = types.ModuleType("DCN")
    m.modulated_deform_conv_forward, m.modulated_deform_conv_backward = fwd, bwd
    sys.modules["DCN"] = m


def load_reference_nlspn():
    _install_stubs()
    _install_dcn_standin()
    pkg = types.ModuleType("refmodel")
    pkg.__path__ = [os.path.join(REF_SRC, "model")]
    sys.modules["refmodel"] = pkg
    for name in ("common", "modulated_deform_conv_func", "nlspnmodel"):
        spec = importlib.util.spec_from_file_location("refmodel." + name, os.path.join(REF_SRC, "model", name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules["refmodel." + name] = mod
        spec.loader.exec_module(mod)
    return sys.modules["refmodel.nlspnmodel"].NLSPN, sys.modules["refmodel.modulated_deform_conv_func"].ModulatedDeformConvFunction


def nlspn_case(NLSPN, name, seed, B, H, W, ch_g=8, k_f=3, prop_time=18, affinity="TGASS", gamma=0.5, conf_prop=True,
               preserve_input=False, legacy=False, with_grad=False, off_scale=0.6, aff_gain=0.5, aff_bias=0.6):
    args = types.SimpleNamespace(prop_time=prop_time, affinity=affinity, affinity_gamma=gamma, conf_prop=conf_prop,
                                 preserve_input=preserve_input, legacy=legacy)
    rs = np.random.RandomState(seed)
    net = NLSPN(args, ch_g, 1, 3, k_f)
    num = k_f * k_f - 1
    wt = (rs.standard_normal(net.conv_offset_aff.weight.shape) * off_scale / np.sqrt(ch_g * 9)).astype(np.float32)
    wt[2 * num:] *= aff_gain                  # affinity logits: kept small enough that 18 iterations stay O(input) (negative affinities amplify)
    bs = (0.3 * rs.standard_normal(3 * num)).astype(np.float32)
    bs[2 * num:] += aff_bias                  # mostly positive affinities (a trained NLSPN's regime): the iteration is then close to an average
    with torch.no_grad():
        net.conv_offset_aff.weight.copy_(torch.from_numpy(wt))
        net.conv_offset_aff.bias.copy_(torch.from_numpy(bs))
    guide = (2.0 * rs.standard_normal((B, ch_g, H, W))).astype(np.float32)
    conf = (1.0 / (1.0 + np.exp(-2.0 * rs.standard_normal((B, 1, H, W))))).astype(np.float32)
    feat = (10.0 * np.abs(rs.standard_normal((B, 1, H, W)))).astype(np.float32)
    fix = (np.where(rs.uniform(size=(B, 1, H, W)) < 0.15, 5.0 + 20.0 * rs.uniform(size=(B, 1, H, W)), 0.0)).astype(np.float32)
    tg, tc, tf, tx = (torch.from_numpy(a) for a in (guide, conf, feat, fix))
    if with_grad:
        tg.requires_grad_(True), tc.requires_grad_(True), tf.requires_grad_(True)
    y, y_inter, offset, aff, aff_const = net(tf, tg, tc if conf_prop else None, tx)
    out = dict(conv_weight=wt, conv_bias=bs, guidance=guide, confidence=conf, feat_init=feat, feat_fix=fix,
               offset_aff=net.conv_offset_aff(tg).detach().numpy(), y=y.detach().numpy(),
               y_inter=np.stack([t.detach().numpy() for t in y_inter]), offset=offset.detach().numpy(), aff=aff.detach().numpy(),
               aff_const=aff_const.numpy(),
               meta=np.array([B, H, W, ch_g, k_f, prop_time, int(conf_prop), int(preserve_input), int(legacy)]), affinity=np.array(affinity))
    if with_grad:
        gy = rs.standard_normal(y.shape).astype(np.float32)
        (y * torch.from_numpy(gy)).sum().backward()
        out.update(grad_y=gy, g_feat_init=tf.grad.numpy(), g_guidance=tg.grad.numpy(), g_confidence=tc.grad.numpy(),
                   g_conv_weight=net.conv_offset_aff.weight.grad.numpy(), g_conv_bias=net.conv_offset_aff.bias.grad.numpy())
        if net.aff_scale_const.grad is not None:
            out["g_aff_scale_const"] = net.aff_scale_const.grad.numpy()
    np.savez_compressed(os.path.join(HERE, f"nlspn_{name}.npz"), **out)
    print(f"nlspn_{name}: |y| max {np.abs(out['y']).max():.3f}, |offset| max {np.abs(out['offset']).max():.2f}, "
          f"aff in [{out['aff'].min():.3f}, {out['aff'].max():.3f}]")


def dcn_case(name, seed, B, C, Co, H, W, kh, kw, stride, pad, dil, group, dg, step):
    rs = np.random.RandomState(seed)
    x = rs.standard_normal((B, C, H, W)).astype(np.float32)
    w = rs.standard_normal((Co, C // group, kh, kw)).astype(np.float32)
    b = rs.standard_normal(Co).astype(np.float32)
    Ho, Wo = dcn_ref.out_size(H, W, kh, kw, stride, pad, dil)
    off = (2.0 * rs.standard_normal((B, dg * 2 * kh * kw, Ho, Wo))).astype(np.float32)
    m = rs.uniform(0, 2, (B, dg * kh * kw, Ho, Wo)).astype(np.float32)
    go = rs.standard_normal((B, Co, Ho, Wo)).astype(np.float32)
    y = dcn_ref.forward(x, w, b, off, m, stride, pad, dil, group, dg, step)
    gi, goff, gm, gw, gb = dcn_ref.backward(x, w, b, off, m, go, stride, pad, dil, group, dg, step)
    np.savez_compressed(os.path.join(HERE, f"dcn_{name}.npz"), input=x, weight=w, bias=b, offset=off, mask=m, grad_out=go, out=y,
                        g_input=gi, g_offset=goff, g_mask=gm, g_weight=gw, g_bias=gb,
                        meta=np.array([stride[0], stride[1], pad[0], pad[1], dil[0], dil[1], group, dg, step]))
    print(f"dcn_{name}: out {y.shape}")


def main():
    NLSPN, _ = load_reference_nlspn()
    torch.manual_seed(0)
    nlspn_case(NLSPN, "tgass", 11, 2, 20, 28, with_grad=True)
    nlspn_case(NLSPN, "preserve", 12, 1, 17, 23, prop_time=6, affinity="ASS", preserve_input=True, aff_gain=2.0, aff_bias=0.0)
    nlspn_case(NLSPN, "as_noconf", 13, 2, 9, 14, prop_time=4, affinity="AS", conf_prop=False, aff_gain=2.0, aff_bias=0.0)
    nlspn_case(NLSPN, "tc_legacy", 14, 1, 12, 10, prop_time=3, affinity="TC", legacy=True)
    nlspn_case(NLSPN, "k5", 15, 1, 14, 18, k_f=5, prop_time=2)
    dcn_case("groups", 21, 2, 4, 6, 7, 9, 3, 3, (1, 1), (1, 1), (1, 1), 2, 1, 1)
    dcn_case("dg_stride", 22, 2, 4, 4, 9, 8, 3, 3, (2, 1), (1, 1), (1, 2), 1, 2, 2)
    dcn_case("k1", 23, 3, 2, 3, 6, 5, 1, 1, (1, 1), (0, 0), (1, 1), 1, 1, 64)


if __name__ == "__main__":
    main()
