#!/usr/bin/env python3
"""One library variant (DDEPTH_LIBRARY): quick parity against the fp64 oracle on ragged shapes, then loop / per-layer times at KITTI size.
    DDEPTH_LIBRARY=build_variants/libddepth_x.so python tools/variant_bench.py [B ...]"""
import os, sys, json
os.environ.setdefault("DDEPTH_STREAMS", "1")      # kernel-level measurements: one stream (the binding defaults to two concurrent lanes)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np, torch
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth
from oracle import ddim_oracle as O

name = os.path.basename(os.environ.get("DDEPTH_LIBRARY", "default"))
sd = synth.make_state_dict(7240)
be = dda.HipDenoiser(); be.load_state_dict(sd); be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
# with DD_CMP=1: the full-size result under the library's default options, to compare the optioned run against (same process, same inputs)
cmp_ref = None
if os.environ.get("DD_CMP"):
    _i = synth.make_inputs(7240, 4, 176, 608)
    _prec = os.environ.get("DD_PRECS", "bf16").split(",")[0]
    cmp_ref = be.denoise(torch.from_numpy(_i["x_T"]).cuda(), torch.from_numpy(_i["cond"]).cuda(), 20, _prec).cpu().numpy()
for kv in filter(None, os.environ.get("DD_OPTS", "").split(",")):      # e.g. DD_OPTS=gn_table=0,hoist_cond=0
    k, v = kv.split("="); be.set_option(k, int(v)); name += f" {k}={v}"
if cmp_ref is not None:
    got = be.denoise(torch.from_numpy(_i["x_T"]).cuda(), torch.from_numpy(_i["cond"]).cuda(), 20, _prec).cpu().numpy()
    print(f"[{name}] KITTI B=4 {_prec} vs default options: max |diff| / max |x_0| = {float(np.abs(got - cmp_ref).max()) / float(np.abs(cmp_ref).max()):.3e}, finite {bool(np.isfinite(got).all())}", flush=True)
ok = True
for (B, h, w, T) in [(2, 9, 33, 3), (1, 24, 40, 5), (2, 17, 70, 2)]:
    inp = synth.make_inputs(50 + h, B, h, w)
    ref = O.ddim_loop(sd, inp["x_T"], inp["cond"], T)
    x, c = torch.from_numpy(inp["x_T"]).cuda(), torch.from_numpy(inp["cond"]).cuda()
    for prec, tol in (("fp32", 2e-5), ("bf16", 1e-2), ("f16", 1.5e-3), ("f16r", 8e-4)):
        e = float(np.abs(be.denoise(x, c, T, prec).cpu().numpy() - ref).max()) / float(np.abs(ref).max())
        if not e < tol:
            ok = False
            print(f"[{name}] PARITY FAIL {B}x{h}x{w} T={T} {prec}: rel err {e:.3e}")
print(f"[{name}] parity {'ok' if ok else 'FAILED'}")
h, w, T = 176, 608, 20
for B in [int(v) for v in sys.argv[1:]] or [4, 1]:
    inp = synth.make_inputs(7240, B, h, w)
    x, c = torch.from_numpy(inp["x_T"]).cuda(), torch.from_numpy(inp["cond"]).cuda()
    for prec in os.environ.get("DD_PRECS", "bf16").split(","):
        be.set_option("timing", 1)
        lm = []
        for _ in range(6):
            be.denoise(x, c, T, prec); lm.append(be.last_loop_ms())
        be.set_option("timing", 0)
        be.set_option("layer_timing", 1)
        for _ in range(2):
            be.denoise(x, c, T, prec)
        torch.cuda.synchronize()
        per = {}
        for l in (1, 2, 3, 9, 4):
            ms, n = be.layer_ms(l)
            if n:
                per[l] = round(ms / n * 1e3, 1)
        be.set_option("layer_timing", 0)
        loop = sorted(lm)[len(lm) // 2]
        # the same loop as two concurrent lanes (the shipped default): wall clock over 8 calls
        two = ""
        if B >= 2:
            be.set_option("streams", 2)
            for _ in range(2):
                be.denoise(x, c, T, prec)
            torch.cuda.synchronize()
            import time
            t0 = time.perf_counter()
            for _ in range(8):
                be.denoise(x, c, T, prec)
            torch.cuda.synchronize()
            two = f", two lanes {B * 8 / (time.perf_counter() - t0):.1f} maps/s"
            be.set_option("streams", 1)
        print(f"[{name}] B={B} {prec}: loop {loop:.3f} ms = {B / loop * 1e3:.1f} maps/s (loop only, one stream){two}, per-layer us {per}", flush=True)
