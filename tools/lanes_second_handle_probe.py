#!/usr/bin/env python3
"""Does a handle's two-lane loop slow down after ANOTHER handle of the same process has run lanes / a B=1 plan / a training pass?
(bench.py's default run once showed a head forward on a second handle at 14.8 instead of 9.2 ms.)  Times handle B's two-lane B=4 loop
fresh, then after each kind of activity on handle A."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ.setdefault("DDEPTH_STREAMS", "1")
import torch
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth

h, w, T = 176, 608, 20
sd = synth.make_state_dict(7240)
inp = synth.make_inputs(7240, 4, h, w)
x, c = torch.from_numpy(inp["x_T"]).cuda(), torch.from_numpy(inp["cond"]).cuda()
g = torch.randn_like(x[:1])


def make():
    be = dda.HipDenoiser(); be.load_state_dict(sd); be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
    return be


def time_lanes(be, S, n=10):
    be.set_option("streams", S)
    for _ in range(3):
        be.denoise(x, c, T, "bf16")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        be.denoise(x, c, T, "bf16")
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


A, Bh = make(), make()
print(f"handle B fresh: one stream {time_lanes(Bh, 1):.3f} ms, two lanes {time_lanes(Bh, 2):.3f} ms", flush=True)
print(f"handle A: one stream {time_lanes(A, 1):.3f} ms, two lanes {time_lanes(A, 2):.3f} ms", flush=True)
print(f"handle B after A ran lanes: two lanes {time_lanes(Bh, 2):.3f} ms", flush=True)
A.set_option("streams", 1)
for _ in range(3):
    A.denoise(x[:1].contiguous(), c[:1].contiguous(), T, "bf16")
print(f"handle B after A ran a B=1 plan: two lanes {time_lanes(Bh, 2):.3f} ms", flush=True)
for _ in range(3):
    A.zero_grad(); A.denoise(x[:1].contiguous(), c[:1].contiguous(), T, "bf16", keep_trajectory=True)
    A.denoise_backward(x[:1].contiguous(), c[:1].contiguous(), g, T, "bf16", trajectory_ticket=A.last_trajectory_ticket)
torch.cuda.synchronize()
print(f"handle B after A ran a training pass: two lanes {time_lanes(Bh, 2):.3f} ms, one stream {time_lanes(Bh, 1):.3f} ms", flush=True)
C = make()
print(f"a third handle created now: one stream {time_lanes(C, 1):.3f} ms, two lanes {time_lanes(C, 2):.3f} ms", flush=True)
