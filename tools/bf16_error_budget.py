#!/usr/bin/env python3
"""CPU experiment for DESIGN.md section 4: where does the depth error of the bf16 mode come from?

Emulates the Res denoiser loop with the fused kernels' rounding points (fp32 state, accumulators and GroupNorm statistics; conv outputs
y1..y3 STORED in 16 bit; MFMA operands = weights and normalised activations rounded to 16 bit; conv4's output fp32) and switches the
rounding sources on one at a time (or all but one), reporting the decoded-depth RMSE against the fp32 loop.  Sources:

    x   state x (fp32) -> operand of conv1            w1..w4  the four convolutions' weights
    s1..s3  storage of y1, y2, y3                     a1, f, a3  operands of conv2, conv3, conv4 (after GroupNorm + ReLU [+ cond + E])
    c   storage of the condition map

    python tools/bf16_error_budget.py [--h 44 --w 152 --T 20 --seeds 2] [--only x,w1 | --plan name=src+src:dtype,...]

Test infrastructure (imports oracle/); nothing here is part of the product."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffusiondepth_amd import synth  # noqa: E402
from oracle import torch_cpu_port as P  # noqa: E402

SOURCES = ["x", "w1", "w2", "w3", "w4", "s1", "s2", "s3", "a1", "f", "a3", "c"]
DT = {"bf16": torch.bfloat16, "f16": torch.float16, "fp32": None}


def split2(x):
    """hi + lo bf16 pair (what an error-compensated operand would carry): value accurate to ~16 mantissa bits"""
    hi = x.to(torch.bfloat16).float()
    lo = (x - hi).to(torch.bfloat16).float()
    return hi + lo


def rnd(x, dt):
    if dt is None:
        return x
    if dt == "bf16x2":
        return split2(x)
    if dt == "q15p":           # int16 with one fp32 scale per PIXEL (max |.| over its channels): what y3 travels as in the refined f16 mode
        sc = x.abs().amax(dim=1, keepdim=True).clamp_min(1e-30) / 32767.0
        return torch.round(x / sc) * sc
    if dt == "q15b":           # int16 with one scale per 32-pixel x 32-channel block (a wave's accumulator block): the hoisted conv3(cond) term
        B, C, h, w = x.shape
        wp = (w + 31) // 32 * 32
        xp = F.pad(x, (0, wp - w))
        blk = xp.view(B, C // 32, 32, h, wp // 32, 32)
        sc = blk.abs().amax(dim=(2, 5), keepdim=True).clamp_min(1e-30) / 32767.0
        return (torch.round(blk / sc) * sc).view(B, C, h, wp)[..., :w]
    if dt == "f16x2":          # hi + lo f16 pair (the split-f16 mode's operands: ~22 mantissa bits)
        hi = x.to(torch.float16).float()
        return hi + (x - hi).to(torch.float16).float()
    return x.to(dt).float()


def denoiser_emul(sd, x, t, cond, R, step=0):
    """R: source -> dtype (None = exact).  'sr' in R: weights stochastically rounded with a per-step seed."""
    emb = F.embedding(torch.as_tensor(t, dtype=torch.long), sd["model.time_embedding.weight"])[..., None, None]

    def W(name, key, li):
        w = sd[name + ".weight"]
        dt = R.get(key)
        if dt is not None and R.get("sr"):
            # stochastic rounding to bf16, a different draw per DDIM step: E[w~] = w, errors of different steps independent
            g = torch.Generator().manual_seed(1000 * li + step)
            u = w.view(torch.int32)
            r = torch.randint(0, 1 << 16, w.shape, generator=g, dtype=torch.int32)
            return ((u + r) & ~0xFFFF).view(torch.float32)
        return rnd(w, dt)

    def gn_relu(y, ys, gk):
        B, C = y.shape[:2]
        mean = y.view(B, 4, -1).mean(-1)
        var = y.view(B, 4, -1).var(-1, unbiased=False)
        yn = (ys.view(B, 4, -1) - mean[..., None]) / torch.sqrt(var[..., None] + 1e-5)
        yn = yn.view_as(y) * sd[gk + ".weight"].view(1, C, 1, 1) + sd[gk + ".bias"].view(1, C, 1, 1)
        return F.relu(yn)

    y1 = F.conv2d(rnd(x, R.get("x")), W("model.noise_embedding.0", "w1", 1), sd["model.noise_embedding.0.bias"], padding=1)
    a1 = gn_relu(y1, rnd(y1, R.get("s1")), "model.noise_embedding.1")
    y2 = F.conv2d(rnd(a1, R.get("a1")), W("model.noise_embedding.3", "w2", 2), sd["model.noise_embedding.3.bias"], padding=1)
    a2 = gn_relu(y2, rnd(y2, R.get("s2")), "model.noise_embedding.4")
    if R.get("hoist"):
        # conv3 is linear: conv3(a2 + cond + E) = conv3(a2) + [conv3(cond) + conv3(E)]; the bracket is computed once per image / per
        # step outside the MFMA loop ("hoistc": dtype of cond and of the weights in that once-per-image convolution)
        hc = R.get("hoistc")
        # "hs": storage of the once-per-image term conv3(cond) (the kernels keep it as f16 quads or fp32); the E[t] tap sums are an fp32 table
        # ("hcc" / "hcw": the condition map / the weights of that once-per-image convolution separately; default = "hoistc" for both)
        hterm = rnd(F.conv2d(rnd(cond, R.get("hcc", hc)), rnd(sd["model.pred.0.weight"], R.get("hcw", hc)), None, padding=1), R.get("hs"))
        y3 = F.conv2d(rnd(a2, R.get("f")), W("model.pred.0", "w3", 3), sd["model.pred.0.bias"], padding=1) + hterm + \
            F.conv2d(emb.expand_as(cond).contiguous(), sd["model.pred.0.weight"], None, padding=1)
    else:
        f = a2 + rnd(cond, R.get("c")) + emb
        y3 = F.conv2d(rnd(f, R.get("f")), W("model.pred.0", "w3", 3), sd["model.pred.0.bias"], padding=1)
    a3 = gn_relu(y3, rnd(y3, R.get("s3")), "model.pred.1")
    y4 = F.conv2d(rnd(a3, R.get("a3")), W("model.pred.3", "w4", 4), sd["model.pred.3.bias"], padding=1)
    return F.relu(F.group_norm(y4, 4, sd["model.pred.4.weight"], sd["model.pred.4.bias"]))


@torch.no_grad()
def loop(sd, x_T, cond, T, R):
    acp = P.make_alphas_cumprod(1000)
    x = torch.as_tensor(x_T)
    cond = torch.as_tensor(cond)
    for k, t in enumerate(P.timesteps(T, 1000)):
        eps = denoiser_emul(sd, x, int(t), cond, R, k)
        x = P.ddim_step(acp, eps, int(t), x, 1000 // T)
    return x


def parse_plan(spec):
    """'name=src+src:dtype/src:dtype'  -> (name, {src: dtype})"""
    name, body = spec.split("=", 1)
    R = {}
    for part in body.split("/"):
        if part in ("sr", "hoist"):
            R[part] = True
            continue
        srcs, dt = part.split(":")
        for s in (SOURCES if srcs == "all" else srcs.split("+")):     # plus the pseudo-sources "hoistc", "hs"
            R[s] = dt if dt in ("bf16x2", "f16x2", "q15p", "q15b") else DT[dt]
    return name, R


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=44)
    ap.add_argument("--w", type=int, default=152)
    ap.add_argument("--T", type=int, default=20)
    ap.add_argument("--B", type=int, default=1)
    ap.add_argument("--seeds", type=int, default=2)
    ap.add_argument("--plan", action="append", default=[], help="name=src+src:dtype/src:dtype[/sr]; src 'all' = every source")
    ap.add_argument("--log-scale", type=float, default=0.0, help="synth.make_state_dict(decoder_log_scale=...): depths times e^s")
    ap.add_argument("--sweep", action="store_true", help="each source alone and all-but-one, in bf16")
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    plans = [parse_plan(p) for p in a.plan]
    if a.sweep or not plans:
        plans += [("all bf16", {s: torch.bfloat16 for s in SOURCES}), ("all f16", {s: torch.float16 for s in SOURCES})]
        plans += [(f"only {s}", {s: torch.bfloat16}) for s in SOURCES]
        plans += [(f"all but {s}", {q: torch.bfloat16 for q in SOURCES if q != s}) for s in SOURCES]
    rows = {}
    for sidx in range(a.seeds):
        sd = P.to_torch_sd(synth.make_state_dict(7240 + sidx, decoder_log_scale=a.log_scale))
        inp = synth.make_inputs(100 + sidx, a.B, a.h, a.w)
        ref = P.decode(sd, P.ddim_loop(sd, inp["x_T"], inp["cond"], a.T))
        for name, R in plans:
            d = P.decode(sd, loop(sd, inp["x_T"], inp["cond"], a.T, R))
            e = d - ref
            rows.setdefault(name, []).append((float(e.pow(2).mean().sqrt()), float(e.abs().max())))
        print(f"seed {sidx}: depth range {float(ref.min()):.2f}..{float(ref.max()):.2f}", flush=True)
    print(f"\nlatent {a.B}x{a.h}x{a.w}, T={a.T}: decoded-depth error vs the fp32 loop (mean over {a.seeds} seeds)")
    for k, v in rows.items():
        print(f"  {k:28s} rmse {np.mean([x[0] for x in v]):.3e}   max {np.mean([x[1] for x in v]):.3e}")


if __name__ == "__main__":
    main()
