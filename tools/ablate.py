#!/usr/bin/env python3
"""Timing ablation of the v2 conv kernels (results are WRONG under ablation; timing only).
Prints per-layer average launch time (us) for a set of ablation masks."""
import os, sys, json
os.environ.setdefault("DDEPTH_STREAMS", "1")      # kernel-level measurements: one stream (the binding defaults to two concurrent lanes)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth

prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
h, w, T = 176, 608, 20
sd = synth.make_state_dict(7240)
be = dda.HipDenoiser(); be.load_state_dict(sd); be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
inp = synth.make_inputs(7240, B, h, w)
x, cond = torch.from_numpy(inp["x_T"]).cuda(), torch.from_numpy(inp["cond"]).cuda()
names = {0: "full", 1: "-transform", 3: "-transform-rawload", 4: "-weightDMA", 7: "-all staging", 8: "-MFMA", 16: "-stores", 32: "-stats",
         48: "-stores-stats", 64: "-barrier", 15: "-staging-MFMA", 63: "only prologue+sync", 127: "only prologue", 256: "return at entry", 512: "return after loads", 1024: "return after patch0", 119: "prologue+MFMA+epi-novmem"}
out = {}
for mask in ([int(m) for m in os.environ["ABL_MASKS"].split(",")] if os.environ.get("ABL_MASKS") else [0, 4, 1, 3, 7, 64, 8, 16, 32, 119, 127, 0]):
    be.set_option("ablate", mask)
    be.set_option("layer_timing", 1)
    for _ in range(2):
        be.denoise(x, cond, T, prec)
    torch.cuda.synchronize()
    per = {l: be.layer_ms(l) for l in (1, 2, 3, 9, 4)}
    be.set_option("layer_timing", 0)
    row = {l: round(per[l][0] / max(per[l][1], 1) * 1e3, 2) for l in per}
    print(f"ablate={mask:3d} {names.get(mask, ''):22s} conv1 {row[1]:7.2f}  conv2 {row[2]:7.2f}  conv3 {max(row[3], row[9]):7.2f}  conv4 {row[4]:7.2f}  sum {sum(row.values()):7.2f} us", flush=True)
    out[str(mask)] = row
be.set_option("ablate", 0)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", f"ablate_{prec}_b{B}.json"), "w"), indent=1)
