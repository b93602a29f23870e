#!/usr/bin/env python3
"""bench.py's default run once showed the head forward (second handle, two lanes) at 14.8 instead of 9.2 ms after the bench's own handle
had run lanes, a B=1 plan and a training pass.  Replays that order and prints per-forward times and the head backend's counters.  Finding
(profiles/history/r02_run29_lanes_head_trace.md): ONE forward of the five stalls ~25 ms on the host between two library calls; the others take 8.5 ms."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
os.environ["DDEPTH_STREAMS"] = sys.argv[1] if len(sys.argv) > 1 else "2"
import torch
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth

dev = torch.device("cuda", 0)
H, W, T, B = 352, 1216, 20, 4
h, w = synth.latent_hw(H, W)
sd = synth.make_state_dict(7240)
inp = synth.make_inputs(7240, B, h, w)
x, c = torch.from_numpy(inp["x_T"]).cuda(), torch.from_numpy(inp["cond"]).cuda()
be = dda.HipDenoiser(); be.load_state_dict(sd); be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
pre = sys.argv[2] if len(sys.argv) > 2 else "lanes,b1,train"
be.set_option("streams", 1)
for _ in range(3): be.denoise(x, c, T, "bf16")
if "lanes" in pre:
    be.set_option("streams", 2)
    for _ in range(3): be.denoise(x, c, T, "bf16")
    be.set_option("streams", 1)
if "b1" in pre:
    for _ in range(3): be.denoise(x[:1].contiguous(), c[:1].contiguous(), T, "bf16")
if "train" in pre:
    g = torch.randn_like(x[:1])
    for _ in range(3):
        be.zero_grad(); be.denoise(x[:1].contiguous(), c[:1].contiguous(), T, "bf16", keep_trajectory=True)
        be.denoise_backward(x[:1].contiguous(), c[:1].contiguous(), g, T, "bf16", trajectory_ticket=be.last_trajectory_ticket)
torch.cuda.synchronize()

sdh = dict(sd); sdh.update(synth.make_fpn_state_dict(7241))
head = dda.DDIMDepthEstimate_Res(precision="bf16", condition_backend="hip", inference_steps=T).eval()
head.load_state_dict({k: torch.from_numpy(v) for k, v in sdh.items()}, strict=False)
head = head.to(dev)
fp = [torch.from_numpy(f).to(dev) for f in synth.make_backbone_features(1, B, H, W)]
gt = torch.from_numpy(synth.make_gt_depth(2, B, H, W)).to(dev)


def timed(n=5):
    with torch.no_grad():
        for _ in range(2):
            head(fp, gt, gt > 0, gt_depth_map=gt)
        torch.cuda.synchronize(dev)
        hb = head._bound.backend
        c0 = {k: hb.counter(k) for k in ("graph_launches", "eager_loops", "plans", "lane_calls")}
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            head(fp, gt, gt > 0, gt_depth_map=gt)
            torch.cuda.synchronize(dev)
            ts.append(round((time.perf_counter() - t0) * 1e3, 2))
        return f"mean {sum(ts) / n:.2f} ms, per forward {ts}", {k: hb.counter(k) - v for k, v in c0.items()}


print(f"DDEPTH_STREAMS={os.environ['DDEPTH_STREAMS']} after [{pre}]:", flush=True)
print("  reference eval behaviour", timed(), flush=True)
head.loss_noise_device = "device"
print("  loss noise on device    ", timed(), flush=True)
head.eval_ddim_loss = False
print("  inference only          ", timed(), flush=True)
