"""First GPU contact of the experimental Winograd convB kernel: parity of one denoiser call, then launch times (layer_timing)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth
be = dda.HipDenoiser(variant="swin"); be.load_state_dict(synth.make_state_dict(7240, "swin")); be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
for (B, h, w) in ((1, 16, 24), (2, 13, 21)):
    inp = synth.make_inputs(5, B, h, w, ((h + 1) // 2, (w + 1) // 2))
    x, cond, t = (torch.from_numpy(inp[k]).cuda() for k in ("x_T", "cond", "timesteps"))
    ref = be.denoise_once(x, t, cond, "fp32")
    for prec in ("f16", "bf16"):
        be.set_option("winograd", 0); d = be.denoise_once(x, t, cond, prec)
        rms = lambda a: float((a - ref).pow(2).mean().sqrt())
        for ver in (1, 2, 3):
            be.set_option("winograd", ver); wv = be.denoise_once(x, t, cond, prec); torch.cuda.synchronize()
            print(f"{B}x{h}x{w} {prec}: eps rms err direct {rms(d):.3e} winograd v{ver} {rms(wv):.3e} finite {bool(torch.isfinite(wv).all())}", flush=True)
B, h, w = 4, 176, 608
inp = synth.make_inputs(7, B, h, w, (88, 304))
x, cond = torch.from_numpy(inp["x_T"]).cuda(), torch.from_numpy(inp["cond"]).cuda()
for wino in (0, 1, 2, 3):
    be.set_option("winograd", wino); be.set_option("layer_timing", 1)
    for _ in range(2): be.denoise(x, cond, 5, "f16")
    torch.cuda.synchronize(); ms, n = be.layer_ms(6); print(f"KITTI B=4 f16 convB winograd={wino}: {1e3 * ms / max(n, 1):.1f} us per launch ({n} launches)", flush=True)
    be.set_option("layer_timing", 0)

# options 4 / 5: every large convolution (Res: conv2, conv3; Swin: conv2, convA, convB, pred.0) on the generalised kernel
for variant in ("res", "swin"):
    b2 = dda.HipDenoiser(variant=variant); b2.load_state_dict(synth.make_state_dict(7240, variant)); b2.set_schedule(dda.DDIMScheduler().alphas_cumprod)
    B, h, w = 2, 21, 45
    inp = synth.make_inputs(9, B, h, w, ((h + 1) // 2, (w + 1) // 2) if variant == "swin" else None)
    x, cond, t = (torch.from_numpy(inp[k]).cuda() for k in ("x_T", "cond", "timesteps"))
    ref = b2.denoise_once(x, t, cond, "fp32")
    for prec, opts in (("f16", (0, 4, 5)), ("bf16", (0, 4))):
        for o in opts:
            b2.set_option("winograd", o); e = b2.denoise_once(x, t, cond, prec); torch.cuda.synchronize()
            print(f"{variant} {prec} winograd={o}: eps rms err {float((e - ref).pow(2).mean().sqrt()):.3e}", flush=True)
    B, h, w = 4, 176, 608
    inp = synth.make_inputs(7, B, h, w, (88, 304) if variant == "swin" else None)
    x, cond = torch.from_numpy(inp["x_T"]).cuda(), torch.from_numpy(inp["cond"]).cuda()
    for o, dma in ((0, 0), (4, 0), (5, 0), (4, 1), (5, 1)):
        b2.set_option("winograd_dma", dma); b2.set_option("winograd", o); b2.set_option("timing", 1)
        for _ in range(3): b2.denoise(x, cond, 20, "f16")
        torch.cuda.synchronize(); print(f"{variant} KITTI B=4 T=20 f16 winograd={o} dma={dma}: loop {b2.last_loop_ms():.2f} ms", flush=True)
