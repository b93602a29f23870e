#!/usr/bin/env python3
"""Latent encoder / decoder kernel times at KITTI size (dd_encode / dd_decode), B = 4 and 1, and their parity against the torch-CPU port.
    python tools/codec_timing.py        (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth
from oracle import torch_cpu_port as P

sd = synth.make_state_dict(7240)
be = dda.HipDenoiser(); be.load_state_dict(sd)
H, W = 352, 1216
h, w = synth.latent_hw(H, W)
for B in (4, 1):
    gt = torch.from_numpy(synth.make_gt_depth(3, B, H, W)).cuda()
    z = (torch.randn(B, 16, h, w, generator=torch.Generator().manual_seed(1)) * 3).cuda()
    for name, fn in (("encode", lambda: be.encode(gt)), ("decode", lambda: be.decode(z))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            out = fn()
        e1.record(); torch.cuda.synchronize()
        print(f"B={B} {name}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
    sdt = P.to_torch_sd(sd)
    d_ref = P.decode(sdt, z[:1].cpu()).numpy()
    d = be.decode(z[:1]).cpu().numpy()
    print(f"B={B} decode max rel err vs torch-CPU port: {float((np.abs(d - d_ref) / np.maximum(np.abs(d_ref), 1e-2)).max()):.2e}")
