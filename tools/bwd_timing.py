#!/usr/bin/env python3
"""Timing of one denoiser call forward (dd_denoise_once) and backward (dd_denoise_once_backward = forward recompute + VJP)
at KITTI latent size.   python tools/bwd_timing.py [batch] [precision] [iters]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 10
h, w = 176, 608
be = dda.HipDenoiser()
be.load_state_dict(synth.make_state_dict(7240))
be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
inp = synth.make_inputs(1, B, h, w)
x, cond = torch.from_numpy(inp["x_T"]).cuda(), torch.from_numpy(inp["cond"]).cuda()
t = torch.randint(0, 1000, (B,), device="cuda")
g = torch.randn_like(x)

def timeit(fn, n):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

fwd = timeit(lambda: be.denoise_once(x, t, cond, prec), iters)
bwd = timeit(lambda: be.denoise_once_backward(x, t, cond, g, prec), iters)
flops_fwd = 626688.0 * B * h * w
print(f"B={B} {prec}: forward {fwd:.3f} ms ({flops_fwd / fwd / 1e9:.0f} TFLOP/s) | backward incl. forward recompute {bwd:.3f} ms "
      f"(VJP alone {bwd - fwd:.3f} ms = {2 * flops_fwd / max(bwd - fwd, 1e-9) / 1e9:.0f} TFLOP/s on the 2x forward FLOPs of dgrad + wgrad)")
