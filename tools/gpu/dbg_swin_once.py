import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth
from oracle import ddim_oracle as O
sd = synth.make_state_dict(7240, "swin")
be = dda.HipDenoiser(variant="swin"); be.load_state_dict(sd); be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
for (B, h, w, ch, cw, tt) in [(1, 9, 33, 5, 17, [37]), (2, 9, 33, 5, 17, [37, 37]), (2, 9, 33, 5, 17, [37, 950]), (3, 9, 33, 5, 17, [37, 950, 512]), (3, 9, 33, 5, 17, [37, 37, 37]), (2, 5, 17, 3, 5, [500, 33]), (2, 24, 40, 12, 20, [37, 950])]:
    i = synth.make_inputs(410 + h, B, h, w, (ch, cw))
    t = np.array(tt, dtype=np.int64)
    ref = O.denoiser_forward(sd, i["x_T"], t, i["cond"], "swin")
    x, c, td = torch.from_numpy(i["x_T"]).cuda(), torch.from_numpy(i["cond"]).cuda(), torch.from_numpy(t).cuda()
    out = {p: be.denoise_once(x, td, c, p).cpu().numpy() for p in ("fp32", "f16", "f16r")}
    print(B, h, w, tt, {p: [round(float(np.abs(out[p][b] - ref[b]).max()), 4) for b in range(B)] for p in out}, flush=True)
