#!/usr/bin/env python3
"""Parity of the precision modes on TRAINED-LIKE weights (every parity figure so far is on seeded random weights whose loop
amplifies |x_0| to ~5e2; no trained checkpoint exists offline).  There is no network for a real checkpoint, so this trains one: the drop-in Res head
(FPN + latent codec in PyTorch-ROCm, denoiser + T-step loop forward AND backward in the library) for N Adam steps on a synthetic task whose
backbone features CARRY the depth (a frozen random strided-conv "backbone" over log-depth, plus noise), then -- on held-out samples, with the
trained parameters -- compares every mode's depth with the reference's own CPU classes (oracle/reference_path.py: CNNDDIMPipiline + inv_t, the
object bench.py's parity gate uses) on the same x_T / condition map.

    python tools/trained_like_parity.py [steps 400] [train precision f16x3] [size kitti|nyu] [batch 2]
prints one JSON line per stage; the last holds, per mode, depth RMSE / max-abs against the reference at the TRAINED decoder's own depth range.
Test infrastructure / experiment: the oracle side is used as the checker only."""
import json
import math
import os
import sys
import time

os.environ.setdefault("DDEPTH_DEVICE_WEIGHTS", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.nn.functional as F

import diffusiondepth_amd as dda
from diffusiondepth_amd import synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
train_prec = sys.argv[2] if len(sys.argv) > 2 else "f16x3"
size = sys.argv[3] if len(sys.argv) > 3 else "kitti"
B = int(sys.argv[4]) if len(sys.argv) > 4 else 2
H, W = {"kitti": (352, 1216), "nyu": (228, 304)}[size]
T = 20
dev = torch.device("cuda", 0)
chans = (64, 128, 256, 512)
gen = torch.Generator(device=dev); gen.manual_seed(int(os.environ.get("TL_SEED", "1234")))
LO, HI, MU, SIGMA = (1.0, 80.0, 1.9, 1.1) if size == "kitti" else (0.5, 10.0, 1.0, 0.6)        # depth range of the synthetic task [m], log-normal body


def depth_batch(n):
    """smooth random depth maps in the data set's range (KITTI 1 .. 80 m, NYU 0.5 .. 10 m): exp of low-pass noise at three scales, 30 % of the pixels without ground truth"""
    z = 0
    for s, a in ((8, 1.0), (24, 0.5), (64, 0.25)):
        z = z + a * F.interpolate(torch.randn((n, 1, max(2, H // (4 * s) + 1), max(2, W // (4 * s) + 1)), device=dev, generator=gen), size=(H, W), mode="bicubic", align_corners=False)
    d = torch.exp(MU + SIGMA * z / 1.15).clamp(LO, HI)
    mask = torch.rand((n, 1, H, W), device=dev, generator=gen) > 0.3
    return d, d * mask


class FrozenBackbone(torch.nn.Module):
    """stand-in for the visual backbone: strided 3x3 convolutions over (log depth, a texture channel) -> 4 non-negative maps at strides 2 / 4 / 8 / 16"""

    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(99)
        cin = 2
        self.convs = torch.nn.ModuleList()
        for c in chans:
            conv = torch.nn.Conv2d(cin, c, 3, 2, 1)
            with torch.no_grad():
                conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * math.sqrt(2.0 / (cin * 9)))
                conv.bias.copy_(torch.randn(conv.bias.shape, generator=g) * 0.1)
            self.convs.append(conv)
            cin = c
        for p in self.parameters():
            p.requires_grad_(False)

    def forward(self, depth):
        x = torch.cat([torch.log(depth) - MU, 0.3 * torch.randn(depth.shape, device=depth.device, generator=gen)], 1)
        out = []
        for conv in self.convs:
            x = F.relu(conv(x))
            out.append(x)
        return out


backbone = FrozenBackbone().to(dev)
head = dda.DDIMDepthEstimate_Res(precision=train_prec, inference_steps=T, loss_noise_device="device")
sd0 = synth.make_state_dict(7240, "res"); sd0.update(synth.make_fpn_state_dict(7241))
head.load_state_dict({k: torch.from_numpy(v) for k, v in sd0.items()}, strict=False)
head = head.to(dev).train()
params = [p for p in head.parameters() if p.requires_grad]
opt = torch.optim.Adam(params, lr=float(os.environ.get("TL_LR", "5e-4")))
t0 = time.perf_counter()
hist = []
for it in range(steps):
    full, gt = depth_batch(B)
    fp = backbone(full)
    opt.zero_grad(set_to_none=True)
    out = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=True)
    m = gt > 0
    l_depth = ((out["pred"] - gt).abs() * m).sum() / m.sum()
    loss = l_depth + out["ddim_loss"]
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 10.0)
    opt.step()
    if it % max(1, steps // 10) == 0 or it == steps - 1:
        hist.append((it, round(float(l_depth.detach()), 4), round(float(out["ddim_loss"].detach()), 4)))
torch.cuda.synchronize()
print(json.dumps({"stage": "train", "steps": steps, "precision": train_prec, "size": size, "batch": B, "seconds": round(time.perf_counter() - t0, 1),
                  "(step, L1 depth loss on valid pixels [m], ddim_loss)": hist}), flush=True)

# ---- the trained parameters, as numpy, into a fresh library handle and into the reference's CPU classes -------------------------------------------
head.eval()
sd = {k: v.detach().float().cpu().numpy() for k, v in head.state_dict().items()}
wmax = max(float(np.abs(v).max()) for k, v in sd.items() if k.startswith("model.") and k.endswith(".weight") and v.ndim == 4)
full, gt = depth_batch(1)
with torch.no_grad():
    fp = backbone(full)
    cond = head.aggregate_condition(fp).float()                 # the trained FPN (eval-mode BatchNorm: running statistics), fp32
    pred_eval = head(fp, gt, gt > 0, gt_depth_map=gt)["pred"]
h, w = synth.latent_hw(H, W)
x_T = torch.randn((1, 16, h, w), device=dev, generator=gen)
be = dda.HipDenoiser(dev, "res")
be.load_state_dict({k: v for k, v in sd.items() if k.startswith(("model.", "depth_transform."))})
be.set_schedule(dda.DDIMScheduler().alphas_cumprod)

from oracle import reference_path as RP  # noqa: E402  (the checker)
from oracle import torch_cpu_port as P  # noqa: E402
t1 = time.perf_counter()
if RP.available():
    pipe_ref, codec_ref = RP.build(sd, "res")
    x0_ref, d_ref = RP.ddim_loop_and_decode(pipe_ref, codec_ref, x_T.cpu(), cond.cpu(), T)
    kind = "reference classes (" + RP.kind() + ")"
else:
    sdt = P.to_torch_sd(sd)
    with torch.no_grad():
        x0_ref = P.ddim_loop(sdt, x_T.cpu(), cond.cpu(), T, variant="res"); d_ref = P.decode(sdt, x0_ref)
    kind = "torch-CPU port"
cpu_s = time.perf_counter() - t1
res = {}
for prec in ("fp32", "f16x3", "f16r", "f16", "bf16"):
    try:
        x0 = be.denoise(x_T, cond, T, prec)
        d = be.decode(x0).cpu()
        e = d - d_ref
        res[prec] = {"depth_rmse": float(torch.sqrt(torch.mean(e ** 2))), "depth_maxabs": float(e.abs().max()),
                     "rel_rmse": float(torch.sqrt(torch.mean((e / d_ref.clamp_min(1e-6)) ** 2))),
                     "latent_maxabs": float((x0.cpu() - x0_ref).abs().max())}
    except RuntimeError as ex:
        res[prec] = {"error": str(ex)[:200]}
m = gt > 0
print(json.dumps({"stage": "parity on the trained parameters", "cpu_side": kind, "cpu_seconds": round(cpu_s, 1),
                  "depth_range_m_of_the_cpu_result": [round(float(d_ref.min()), 3), round(float(d_ref.max()), 3)],
                  "rms_depth_m": round(float(torch.sqrt(torch.mean(d_ref ** 2))), 3),
                  "latent_x0_maxabs": round(float(x0_ref.abs().max()), 2),
                  "largest_conv_weight_of_the_denoiser": round(wmax, 3),
                  "held_out_L1_of_the_head_prediction_m": round(float(((pred_eval - gt).abs() * m).sum() / m.sum()), 3),
                  "modes": res}), flush=True)
