#!/usr/bin/env python3
"""Energy view of the DDIM loop: for each ablation mask (library built with -DDD_ABLATE=1) run the loop back to back for
~1.5 s and sample sclk / socket power -> ms per loop, average clock, average power, joules per loop.
    DDEPTH_LIBRARY=build_variants/lib_abl.so python tools/power_ablate.py [precision] [batch] [mask,mask,...]"""
import os, re, subprocess, sys, threading, time
os.environ.setdefault("DDEPTH_STREAMS", "1")      # kernel-level measurements: one stream (the binding defaults to two concurrent lanes)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
masks = [int(m) for m in (sys.argv[3] if len(sys.argv) > 3 else "0").split(",")]
h, w = 176, 608
be = dda.HipDenoiser()
be.load_state_dict(synth.make_state_dict(7240))
be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
inp = synth.make_inputs(1, B, h, w)
x_T, cond = torch.from_numpy(inp["x_T"]).cuda(), torch.from_numpy(inp["cond"]).cuda()

def sample():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    sclk = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out); pw = re.search(r"Power \(W\): ([\d.]+)", out)
    return (int(sclk.group(1)) if sclk else 0, float(pw.group(1)) if pw else 0.0)

for mask in masks:
    be.set_option("ablate", mask)
    be.denoise(x_T, cond, 20, prec); torch.cuda.synchronize()
    stop = [False]; count = [0]
    def run():
        while not stop[0]:
            be.denoise(x_T, cond, 20, prec); count[0] += 1
            if count[0] % 4 == 0: torch.cuda.synchronize()
        torch.cuda.synchronize()
    t = threading.Thread(target=run); t0 = time.time(); t.start()
    time.sleep(0.6)
    ss = [sample() for _ in range(5)]
    stop[0] = True; t.join(); dt = time.time() - t0
    ms = dt / count[0] * 1e3
    clk = sum(s[0] for s in ss) / len(ss); pw = sum(s[1] for s in ss) / len(ss)
    print(f"mask {mask:5d}: {ms:7.3f} ms/loop  sclk {clk:6.0f} MHz  power {pw:6.0f} W  energy {ms * pw * 1e-3:7.3f} J/loop", flush=True)
be.set_option("ablate", 0)
