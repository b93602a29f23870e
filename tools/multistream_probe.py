#!/usr/bin/env python3
"""Do independent sub-batches on separate HIP streams fill each other's tails and kernel boundaries?  The images of a batch are independent
(GroupNorm is per sample), so a batch of B can run as S concurrent loops of B/S images, each on its own stream with its own handle (plans and
activation buffers are per handle).  Prints maps/s of one B-image loop vs S concurrent (B/S)-image loops.
    python tools/multistream_probe.py [B] [precision]"""
import os, sys, time
os.environ.setdefault("DDEPTH_STREAMS", "1")      # the lanes here are the probe's own handles / streams
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
h, w, T = 176, 608, 20
sd = synth.make_state_dict(7240)
inp = synth.make_inputs(7240, B, h, w)
x, c = torch.from_numpy(inp["x_T"]).cuda(), torch.from_numpy(inp["cond"]).cuda()


def make():
    be = dda.HipDenoiser(); be.load_state_dict(sd); be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
    return be


def bench(S, n=12, stagger_us=0.0):
    hs = [make() for _ in range(S)]
    ss = [torch.cuda.Stream() for _ in range(S)]
    per = B // S
    xs = [x[i * per:(i + 1) * per].contiguous() for i in range(S)]
    cs = [c[i * per:(i + 1) * per].contiguous() for i in range(S)]
    outs = [torch.empty_like(v) for v in xs]

    def once():
        for i in range(S):
            with torch.cuda.stream(ss[i]):
                hs[i].denoise(xs[i], cs[i], T, prec, out=outs[i])
    for _ in range(3):
        once()
    torch.cuda.synchronize()
    if stagger_us > 0:          # streams 1.. start late by i * stagger_us; nothing re-aligns them before the final synchronize
        for i in range(1, S):
            with torch.cuda.stream(ss[i]):
                torch.cuda._sleep(int(i * stagger_us * 100))       # wall-clock counter: 100 MHz
    t0 = time.perf_counter()
    for _ in range(n):
        once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    return dt, torch.cat(outs)


ref = None
for S in [s for s in (1, 2, 4, 8) if B % s == 0 and s <= B]:
    dt, out = bench(S)
    if ref is None:
        ref = out
    err = float((out - ref).abs().max() / ref.abs().max())
    print(f"B={B} {prec}: {S} stream(s) x {B // S} image(s): {dt * 1e3:.3f} ms per {B} maps = {B / dt:.1f} maps/s   (max rel diff vs one stream {err:.1e})", flush=True)

if len(sys.argv) > 3:
    for st in [float(v) for v in sys.argv[3].split(",")]:
        dt, _ = bench(2, n=24, stagger_us=st)
        print(f"B={B} {prec}: 2 streams, second one starts {st:.0f} us late: {dt * 1e3:.3f} ms per {B} maps = {B / dt:.1f} maps/s", flush=True)
