#!/usr/bin/env python3
"""Graph-mode loop time under ablation masks (timing only)."""
import os, sys
os.environ.setdefault("DDEPTH_STREAMS", "1")      # kernel-level measurements: one stream (the binding defaults to two concurrent lanes)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
masks = [int(a) for a in sys.argv[2:]] or [0, 256, 2048, 0, 2048]
h, w, T = 176, 608, 20
be = dda.HipDenoiser(); be.load_state_dict(synth.make_state_dict(7240)); be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
inp = synth.make_inputs(7240, B, h, w)
x, cond = torch.from_numpy(inp["x_T"]).cuda(), torch.from_numpy(inp["cond"]).cuda()
be.set_option("timing", 1)
for m in masks:
    be.set_option("ablate", m)
    ts = []
    for i in range(6):
        be.denoise(x, cond, T, "bf16"); ts.append(be.last_loop_ms())
    ts = sorted(ts[1:])
    print(f"B={B} ablate={m:5d} loop ms median {ts[len(ts)//2]:.4f} min {ts[0]:.4f}  -> {1e3*ts[len(ts)//2]/T:.1f} us/step", flush=True)
