#!/usr/bin/env python3
"""CPU experiment: where does the depth error of the Swin / MPViT denoiser's 16-bit modes come from?  The twin of
tools/bf16_error_budget.py for the UpSample_add variant (reference src/model/head/ddim_depth_estimate_res_swin_addHAHI.py:321-382), emulating the
rounding points of the HOISTED forward-only plans: per step only NE(x_t) crosses 16-bit operands --
    y1 = conv1(x)        a1 = relu(gn1(y1))      y2 = conv2(a1)      a2 = relu(gn2(y2))
    sa = convA'(a2)      [stored, then the operand of the 5x5 form as stored]
    y3 = pred.0(convB(sa)) + H + E-terms          [H = the once-per-image term pred.0(convB(convA(up(feat)) + a) + b)]
    a3 = relu(gn3(y3))   y4 = conv4(a3)          eps = relu(gn4(y4))
Sources:  x w1 s1 a1 w2 s2 a2 wA sA w5 (the composed pred.0 o convB weights: emulated by rounding wB and w3) hc (operands of the once-per-image chain)
hs (storage of H) s3 a3 w4.
    python tools/swin_error_budget.py [--h 44 --w 152 --log-scale 1.8] --plan name=src+src:dtype/...
Test infrastructure (imports oracle/); nothing here is part of the product."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from diffusiondepth_amd import synth  # noqa: E402
from oracle import torch_cpu_port as P  # noqa: E402
from bf16_error_budget import rnd, DT  # noqa: E402

SOURCES = ["x", "w1", "s1", "a1", "w2", "s2", "a2", "wA", "sA", "w5", "hc", "hs", "s3", "a3", "w4"]


def gn_relu(sd, y, ys, gk):
    B, C = y.shape[:2]
    mean = y.view(B, 4, -1).mean(-1)
    var = y.view(B, 4, -1).var(-1, unbiased=False)
    yn = (ys.view(B, 4, -1) - mean[..., None]) / torch.sqrt(var[..., None] + 1e-5)
    return F.relu(yn.view_as(y) * sd[gk + ".weight"].view(1, C, 1, 1) + sd[gk + ".bias"].view(1, C, 1, 1))


def denoiser_emul(sd, x, t, up, R):
    g = R.get
    emb = F.embedding(torch.as_tensor(t, dtype=torch.long), sd["model.time_embedding.weight"])[..., None, None]
    wA, bA = sd["model.upsample_fuse.convA.conv.weight"], sd["model.upsample_fuse.convA.conv.bias"]
    wB, bB = sd["model.upsample_fuse.convB.conv.weight"], sd["model.upsample_fuse.convB.conv.bias"]
    w3, b3 = sd["model.pred.0.weight"], sd["model.pred.0.bias"]
    y1 = F.conv2d(rnd(x, g("x")), rnd(sd["model.noise_embedding.0.weight"], g("w1")), sd["model.noise_embedding.0.bias"], padding=1)
    a1 = gn_relu(sd, y1, rnd(y1, g("s1")), "model.noise_embedding.1")
    y2 = F.conv2d(rnd(a1, g("a1")), rnd(sd["model.noise_embedding.3.weight"], g("w2")), sd["model.noise_embedding.3.bias"], padding=1)
    a2 = gn_relu(sd, y2, rnd(y2, g("s2")), "model.noise_embedding.4")
    # the per-step part: convA' on a2 alone, stored; pred.0 o convB on it (composed 5x5 weights: rounding emulated on both factors)
    sa = rnd(F.conv2d(rnd(a2, g("a2")), rnd(wA, g("wA")), None, padding=1), g("sA"))
    part = F.conv2d(F.conv2d(sa, rnd(wB, g("w5")), None, padding=1), rnd(w3, g("w5")), None, padding=1)
    # the once-per-image term (condition map through all three convolutions, biases of the fuse convolutions) and the E[t] term (fp32 tables)
    hcd = g("hc")
    H = F.conv2d(F.conv2d(F.conv2d(rnd(up, hcd), rnd(wA, hcd), bA, padding=1), rnd(wB, hcd), bB, padding=1), rnd(w3, hcd), None, padding=1)
    E = F.conv2d(F.conv2d(F.conv2d(emb.expand_as(up).contiguous(), wA, None, padding=1), wB, None, padding=1), w3, None, padding=1)
    y3 = part + rnd(H, g("hs")) + E + b3.view(1, -1, 1, 1)
    a3 = gn_relu(sd, y3, rnd(y3, g("s3")), "model.pred.1")
    y4 = F.conv2d(rnd(a3, g("a3")), rnd(sd["model.pred.3.weight"], g("w4")), sd["model.pred.3.bias"], padding=1)
    return F.relu(F.group_norm(y4, 4, sd["model.pred.4.weight"], sd["model.pred.4.bias"]))


@torch.no_grad()
def loop(sd, x_T, cond, T, R):
    acp = P.make_alphas_cumprod(1000)
    x = torch.as_tensor(x_T)
    up = F.interpolate(torch.as_tensor(cond), size=[x.size(2), x.size(3)], mode="bilinear", align_corners=True)
    for t in P.timesteps(T, 1000):
        x = P.ddim_step(acp, denoiser_emul(sd, x, int(t), up, R), int(t), x, 1000 // T)
    return x


def parse_plan(spec):
    name, body = spec.split("=", 1)
    R = {}
    for part in filter(None, body.split("/")):
        srcs, dt = part.split(":")
        for s in (SOURCES if srcs == "all" else srcs.split("+")):
            R[s] = dt if dt in ("bf16x2", "f16x2", "q15p", "q15b") else DT[dt]
    return name, R


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=44)
    ap.add_argument("--w", type=int, default=152)
    ap.add_argument("--T", type=int, default=20)
    ap.add_argument("--seeds", type=int, default=2)
    ap.add_argument("--plan", action="append", default=[])
    ap.add_argument("--log-scale", type=float, default=0.0)
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    plans = [parse_plan(p) for p in a.plan] or [("all f16", {s: torch.float16 for s in SOURCES})] + [(f"only {s}", {s: torch.float16}) for s in SOURCES]
    rows = {}
    for sidx in range(a.seeds):
        sd = P.to_torch_sd(synth.make_state_dict(7240 + sidx, "swin", decoder_log_scale=a.log_scale))
        inp = synth.make_inputs(100 + sidx, 1, a.h, a.w, ((a.h + 1) // 2, (a.w + 1) // 2))
        ref = P.decode(sd, P.ddim_loop(sd, inp["x_T"], inp["cond"], a.T, variant="swin"))
        chk = P.decode(sd, loop(sd, inp["x_T"], inp["cond"], a.T, {}))
        print(f"seed {sidx}: depth range {float(ref.min()):.2f}..{float(ref.max()):.2f}; hoisted order vs reference order (no rounding): {float((chk - ref).abs().max()):.2e}", flush=True)
        for name, R in plans:
            e = P.decode(sd, loop(sd, inp["x_T"], inp["cond"], a.T, R)) - chk
            rows.setdefault(name, []).append((float(e.pow(2).mean().sqrt()), float(e.abs().max())))
    print(f"\nSwin denoiser, latent 1x{a.h}x{a.w}, T={a.T}: decoded-depth error vs the unrounded loop (mean over {a.seeds} seeds)")
    for k, v in rows.items():
        print(f"  {k:28s} rmse {np.mean([x[0] for x in v]):.3e}   max {np.mean([x[1] for x in v]):.3e}")


if __name__ == "__main__":
    main()
