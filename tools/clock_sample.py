#!/usr/bin/env python3
"""Samples sclk / socket power (rocm-smi) while the DDIM loop runs back to back: is the chip clock- or power-limited under
the MFMA kernels?   python tools/clock_sample.py [precision] [batch]"""
import os, re, subprocess, sys, threading, time
os.environ.setdefault("DDEPTH_STREAMS", "1")      # kernel-level measurements: one stream (the binding defaults to two concurrent lanes)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
h, w = 176, 608
be = dda.HipDenoiser()
be.load_state_dict(synth.make_state_dict(7240))
be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
inp = synth.make_inputs(1, B, h, w)
x_T, cond = torch.from_numpy(inp["x_T"]).cuda(), torch.from_numpy(inp["cond"]).cuda()
be.denoise(x_T, cond, 20, prec); torch.cuda.synchronize()
stop = False
count = [0]
def run():
    while not stop:
        be.denoise(x_T, cond, 20, prec); count[0] += 1
        if count[0] % 8 == 0: torch.cuda.synchronize()
    torch.cuda.synchronize()
def sample():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    sclk = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out); pw = re.search(r"Power \(W\): ([\d.]+)", out)
    return (int(sclk.group(1)) if sclk else None, float(pw.group(1)) if pw else None)
print("idle:", sample())
t = threading.Thread(target=run); t0 = time.time(); t.start()
time.sleep(1.0)
for i in range(8):
    print(f"t={time.time() - t0:.1f}s loops={count[0]} sclk/power:", sample(), flush=True)
stop = True; t.join()
dt = time.time() - t0
print(f"{count[0]} loops of B={B} in {dt:.2f}s -> {count[0] * B / dt:.1f} maps/s ({prec})")
