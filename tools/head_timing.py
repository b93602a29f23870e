#!/usr/bin/env python3
"""End-to-end head timing on the GPU: PyTorch-ROCm FPN (condition aggregation) vs the HIP hot path."""
import os, sys, time
os.environ.setdefault("DDEPTH_STREAMS", "1")      # kernel-level measurements: one stream (the binding defaults to two concurrent lanes)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth

H, W, B = 352, 1216, int(sys.argv[1]) if len(sys.argv) > 1 else 1
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
sd = synth.make_state_dict(7240); sd.update(synth.make_fpn_state_dict(7241))
head = dda.DDIMDepthEstimate_Res(precision=prec, condition_backend="torch").eval()
head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
head = head.cuda()
head_hip = dda.DDIMDepthEstimate_Res(precision=prec, condition_backend="hip").eval()
head_hip.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
head_hip = head_hip.cuda()
fp = [torch.from_numpy(f).cuda() for f in synth.make_backbone_features(1, B, H, W)]
gt = torch.from_numpy(synth.make_gt_depth(2, B, H, W)).cuda()

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

with torch.no_grad():
    t_fpn = timeit(lambda: head.aggregate_condition(fp))
    fpcl = [f.contiguous(memory_format=torch.channels_last) for f in fp]
    head_cl = head.to(memory_format=torch.channels_last)
    t_fpn_cl = timeit(lambda: head_cl.aggregate_condition(fpcl))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        t_fpn_bf16 = timeit(lambda: head.aggregate_condition(fp))
    t_head = timeit(lambda: head(fp, gt, gt > 0, gt_depth_map=gt))
    t_fpn_hip = timeit(lambda: head_hip.aggregate_condition(fp))
    be = head_hip._bound.backend
    t_fpn_hip_noexp = timeit(lambda: be.condition(fp, prec, export=False))
    t_head_hip = timeit(lambda: head_hip(fp, gt, gt > 0, gt_depth_map=gt))
    head_hip.loss_noise_device = "device"
    t_head_dev = timeit(lambda: head_hip(fp, gt, gt > 0, gt_depth_map=gt))
    head_hip.eval_ddim_loss = False
    t_head_inf = timeit(lambda: head_hip(fp, gt, gt > 0, gt_depth_map=gt))
    head_hip.eval_ddim_loss, head_hip.loss_noise_device = True, "cpu"
    be.set_option("layer_timing", 1)
    for _ in range(5): be.condition(fp, prec, export=False)
    torch.cuda.synchronize()
    lay = {l: be.layer_ms(l) for l in range(10, 15)}
    be.set_option("layer_timing", 0)
print(f"B={B} {prec}: FPN torch fp32 {t_fpn:.3f} ms | channels_last {t_fpn_cl:.3f} ms | autocast bf16 {t_fpn_bf16:.3f} ms | whole head.forward {t_head:.3f} ms")
print(f"B={B} {prec}: FPN HIP {t_fpn_hip:.3f} ms (no export {t_fpn_hip_noexp:.3f} ms) | whole head.forward with HIP FPN {t_head_hip:.3f} ms")
print(f"B={B} {prec}: whole head.forward, loss noise drawn on the device {t_head_dev:.3f} ms | eval_ddim_loss=False (inference only) {t_head_inf:.3f} ms = {B / t_head_inf * 1e3:.1f} maps/s")
print("   FPN conv kernels avg us:", {l: round(1e3 * ms / max(n, 1), 1) for l, (ms, n) in lay.items()}, "launches", {l: n for l, (ms, n) in lay.items()})
