#!/usr/bin/env python3
"""GPU timing of the NLSPN refinement stage at KITTI 352 x 1216:
  fused path      dd_nlspn_offset_affinity + dd_nlspn_propagate (prop_time launches of nlspn_prop_kernel)
  per-op path     the reference's formulation on the HIP DCNv2 operator: 8 + 18 ModulatedDeformConv forward calls + torch glue
against the HBM roofline: algorithmic bytes per propagation iteration = (18 offset + 9 affinity + 1 result) planes x 4 B = 112 B per
pixel (the gathered depth map, 4 B / pixel, stays in L2), 8 TB/s peak (MI355X_MICROARCH.md).
Usage: python tools/nlspn_timing.py [--batch 1] [--iters 20]   -> one JSON line"""
import argparse
import json
import os
import sys
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from diffusiondepth_amd import dcn  # noqa: E402
from diffusiondepth_amd.nlspn import NLSPN  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--height", type=int, default=352)
    ap.add_argument("--width", type=int, default=1216)
    ap.add_argument("--variants", default="", help="comma-separated DD_NLSPN_KERNEL values to time (g4,g1,g4p,g1p,l8,l16,l32,l8p,...)")
    a = ap.parse_args()
    B, H, W, T = a.batch, a.height, a.width, 18
    args = types.SimpleNamespace(prop_time=T, affinity="TGASS", affinity_gamma=0.5, conf_prop=True, preserve_input=False, legacy=False)
    m = NLSPN(args, 8, 1, 3, 3).cuda().eval()
    gen = torch.Generator(device="cuda").manual_seed(1)
    with torch.no_grad():
        m.conv_offset_aff.weight.copy_(0.1 * torch.randn(m.conv_offset_aff.weight.shape, device="cuda", generator=gen))
        m.conv_offset_aff.bias.copy_(0.3 * torch.randn(24, device="cuda", generator=gen))
        m.conv_offset_aff.bias[16:] += 0.6
    feat = 10 * torch.rand(B, 1, H, W, device="cuda", generator=gen)
    guide = 2 * torch.randn(B, 8, H, W, device="cuda", generator=gen)
    conf = torch.rand(B, 1, H, W, device="cuda", generator=gen)
    with torch.no_grad():
        offset_aff = m.conv_offset_aff(guide)
        offset, aff = dcn.nlspn_offset_affinity(offset_aff, conf, m.aff_scale_const, m.w_conf, m.b, 3, "TGASS", True, False)
        t_conv = timed(lambda: m.conv_offset_aff(guide), a.iters)
        t_aff = timed(lambda: dcn.nlspn_offset_affinity(offset_aff, conf, m.aff_scale_const, m.w_conf, m.b, 3, "TGASS", True, False), a.iters)
        t_prop = timed(lambda: dcn.nlspn_propagate(feat, offset, aff, None, m.w, m.b, 3, T, False), a.iters)
        t_mod = timed(lambda: m(feat, guide, conf), a.iters)
        t_guided = timed(lambda: dcn.nlspn_guided_offset_affinity(guide, m.conv_offset_aff.weight, m.conv_offset_aff.bias, conf, m.aff_scale_const,
                                                                  m.w_conf, m.b, 3, 3, "TGASS", True, False), a.iters)
        m.fuse_guidance_conv = False
        t_mod_unfused = timed(lambda: m(feat, guide, conf), a.iters)
        m.fuse_guidance_conv = True

        def per_op():
            o, af = m._get_offset_affinity(guide, conf)
            f = feat
            for _ in range(T):
                f = m._propagate_once(f, o, af)
            return f
        t_perop = timed(per_op, max(2, a.iters // 4))
        y_f = m(feat, guide, conf)[0]
        y_p = per_op()
    variants = {}
    if a.variants:
        for mode in a.variants.split(","):
            os.environ["DD_NLSPN_KERNEL"] = mode
            with torch.no_grad():
                t = timed(lambda: dcn.nlspn_propagate(feat, offset, aff, None, m.w, m.b, 3, T, False), a.iters)
                y_v = dcn.nlspn_propagate(feat, offset, aff, None, m.w, m.b, 3, T, False)[-1]
            variants[mode] = {"us_per_iter": 1e3 * t / T, "GBps": 112 * B * H * W * T / (t * 1e-3) / 1e9,
                              "maxrel_vs_default": float((y_v - y_f).abs().max() / y_f.abs().max())}
        os.environ.pop("DD_NLSPN_KERNEL", None)
    px = B * H * W
    bytes_iter = 112 * px
    out = {"B": B, "H": H, "W": W, "prop_time": T, "conv_offset_aff_ms": t_conv, "affinity_ms": t_aff, "guided_conv_affinity_ms": t_guided, "module_forward_unfused_conv_ms": t_mod_unfused, "propagate_ms": t_prop,
           "propagate_us_per_iter": 1e3 * t_prop / T, "propagate_GBps": bytes_iter * T / (t_prop * 1e-3) / 1e9,
           "propagate_frac_hbm_peak": bytes_iter * T / (t_prop * 1e-3) / 8e12,
           "affinity_GBps": (24 + 1 + 27) * 4 * px / (t_aff * 1e-3) / 1e9,
           "module_forward_ms": t_mod, "per_op_formulation_ms": t_perop, "maps_per_s": B / (t_mod * 1e-3),
           "fused_vs_per_op_maxrel": float((y_f - y_p).abs().max() / y_p.abs().max()), "variants": variants}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
