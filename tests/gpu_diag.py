#!/usr/bin/env python3
"""GPU diagnostics (not a pytest file): per-layer raw conv outputs of one denoiser call, every precision
mode, against the fp64 oracle.  Writes gpurun_out/diag.txt.  Run on the GPU box:  python tests/gpu_diag.py"""
import os
import sys
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import diffusiondepth_amd as dda  # noqa: E402
from diffusiondepth_amd import synth  # noqa: E402
from oracle import ddim_oracle as O  # noqa: E402


def oracle_layers(sd, x, t, cond):
    sd = {k: v.astype(np.float64) for k, v in sd.items()}
    x = x.astype(np.float64); cond = cond.astype(np.float64)
    B = x.shape[0]
    y1 = O.conv2d(x, sd["model.noise_embedding.0.weight"], sd["model.noise_embedding.0.bias"])
    a1 = O.relu(O.group_norm(y1, 4, sd["model.noise_embedding.1.weight"], sd["model.noise_embedding.1.bias"]))
    y2 = O.conv2d(a1, sd["model.noise_embedding.3.weight"], sd["model.noise_embedding.3.bias"])
    a2 = O.relu(O.group_norm(y2, 4, sd["model.noise_embedding.4.weight"], sd["model.noise_embedding.4.bias"]))
    emb = sd["model.time_embedding.weight"][np.asarray(t)].reshape(B, -1, 1, 1)
    f = cond + emb + a2
    y3 = O.conv2d(f, sd["model.pred.0.weight"], sd["model.pred.0.bias"])
    a3 = O.relu(O.group_norm(y3, 4, sd["model.pred.1.weight"], sd["model.pred.1.bias"]))
    y4 = O.conv2d(a3, sd["model.pred.3.weight"], sd["model.pred.3.bias"])
    eps = O.relu(O.group_norm(y4, 4, sd["model.pred.4.weight"], sd["model.pred.4.bias"]))
    return {"y1": y1, "y2": y2, "y3": y3, "y4": y4, "eps": eps}


def main():
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    lines = []

    def say(s):
        print(s, flush=True)
        lines.append(s)

    say(f"torch {torch.__version__} device {torch.cuda.get_device_name(0)}")
    sd = synth.make_state_dict(7240)
    be = dda.HipDenoiser()
    say(be.version)
    be.load_state_dict(sd)
    be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
    for (B, h, w) in [(2, 12, 20), (1, 19, 45)]:
        inp = synth.make_inputs(11, B, h, w)
        ref = oracle_layers(sd, inp["x_T"], inp["timesteps"], inp["cond"])
        x, cond, t = (torch.from_numpy(inp[k]).cuda() for k in ("x_T", "cond", "timesteps"))
        for prec in ("naive_fp32", "fp32", "bf16", "f16"):
            try:
                eps = be.denoise_once(x, t, cond, prec)
                torch.cuda.synchronize()
                row = [f"B{B} {h}x{w} {prec:10s}"]
                for name in ("y1", "y2", "y3", "y4"):
                    got = be.debug_fetch(name, B, h, w).cpu().numpy()
                    err = np.abs(got - ref[name]).max() / np.abs(ref[name]).max()
                    row.append(f"{name} rel {err:.2e}")
                e = np.abs(eps.cpu().numpy() - ref["eps"]).max()
                row.append(f"eps abs {e:.2e} finite={bool(torch.isfinite(eps).all())}")
                say("  ".join(row))
            except Exception:
                say(f"B{B} {h}x{w} {prec}: EXCEPTION\n" + traceback.format_exc())
    # loop, graph on/off
    inp = synth.make_inputs(1, 1, 24, 40)
    x, cond = torch.from_numpy(inp["x_T"]).cuda(), torch.from_numpy(inp["cond"]).cuda()
    ref = O.ddim_loop(sd, inp["x_T"], inp["cond"], 20)
    for prec in ("naive_fp32", "fp32", "bf16", "f16"):
        for graph in (0, 1):
            try:
                be.set_option("graph", graph)
                x0 = be.denoise(x, cond, 20, prec).cpu().numpy()
                say(f"loop T=20 24x40 {prec:10s} graph={graph}: latent maxabs {np.abs(x0 - ref).max():.3e} "
                    f"(scale {np.abs(ref).max():.1f}) graph_launches={be.counter('graph_launches')} "
                    f"capture_failures={be.counter('graph_capture_failures')}")
            except Exception:
                say(f"loop {prec} graph={graph}: EXCEPTION\n" + traceback.format_exc())
    with open(os.path.join(ROOT, "gpurun_out", "diag.txt"), "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()
