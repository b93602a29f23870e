#!/usr/bin/env python3
"""Round 6: the default line is 23 % slower under torch.distributed.run with ONE rank (profiles/r06_call32_launcher_probe.txt).  Is it the launcher's environment or the
RCCL process group itself?  One process, no launcher: the KITTI B = 4 f16r step (encode + 20-step loop on two lanes + decode) timed with events
  (1) before any process group exists, (2) after init_process_group("nccl", world 1) [+ one barrier], (3) after destroy_process_group,
each with the per-layer launch times of an eager one-stream pass.  argv: [device_id 0|1 = pass device_id to init_process_group (eager communicator)]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth

eager = (sys.argv[1] == "1") if len(sys.argv) > 1 else True
pg_first = len(sys.argv) > 2 and sys.argv[2] == "first"         # the process group BEFORE the library handle exists (bench.py's order under a launcher)
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
if pg_first:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev if eager else None)
    dist.barrier(); torch.cuda.synchronize()
H, W, B, T = 352, 1216, 4, 20
h, w = synth.latent_hw(H, W)
be = dda.HipDenoiser(dev); be.load_state_dict(synth.make_state_dict(7240)); be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
be.set_option("streams", 2)
if os.environ.get("LANE_PROBE") is not None:
    be.set_option("lane_probe", int(os.environ["LANE_PROBE"]))
inp = synth.make_inputs(7240, B, h, w)
x_T, cond = torch.from_numpy(inp["x_T"]).to(dev), torch.from_numpy(inp["cond"]).to(dev)
gt = torch.from_numpy(synth.make_gt_depth(7240, B, H, W)).to(dev)
x0 = torch.empty_like(x_T)


def step():
    be.encode(gt); be.denoise(x_T, cond, T, "f16r", out=x0); return be.decode(x0)


def measure(tag):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        step()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    be.set_option("layer_timing", 1)
    for _ in range(2):
        be.denoise(x_T, cond, T, "f16r", out=x0)
    torch.cuda.synchronize()
    per = {l: be.layer_ms(l) for l in (1, 2, 9, 4)}
    be.set_option("layer_timing", 0)
    print(f"{tag}: step {ms:.3f} ms = {B / ms * 1e3:.1f} maps/s; per-layer us " + str({l: round(v[0] / max(v[1], 1) * 1e3, 1) for l, v in per.items()}) +
          f"; lane_overlap {be.counter('lane_overlap')}, lane_probe_retries {be.counter('lane_probe_retries')}", flush=True)


measure("process group created FIRST (+ one barrier), then the handle" if pg_first else "no process group")
if pg_first:
    dist.destroy_process_group()
    measure("after destroy_process_group")
    sys.exit(0)
if eager:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
else:
    dist.init_process_group("nccl", rank=0, world_size=1)
measure(f"after init_process_group(nccl, world 1, eager communicator = {eager})")
dist.barrier(); torch.cuda.synchronize()
measure("after one barrier")
t = torch.ones(1, device=dev); dist.all_reduce(t); torch.cuda.synchronize()
measure("after one all_reduce")
dist.destroy_process_group()
measure("after destroy_process_group")
