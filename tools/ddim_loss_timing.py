#!/usr/bin/env python3
"""Where do the 4.9 ms of the head's every-forward ddim_loss go?  Times the stages of
DDIMDepthEstimate_Res.ddim_loss one by one at KITTI size (hipEvents around each stage, median of 7), with the loss noise on the device.
    python tools/ddim_loss_timing.py [batch] [precision]
Run it under `rocprofv3 --kernel-trace --stats` for the kernel view."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16"
H, W = 352, 1216
sd = synth.make_state_dict(7240); sd.update(synth.make_fpn_state_dict(7241))
head = dda.DDIMDepthEstimate_Res(precision=prec, condition_backend="hip", inference_steps=20, loss_noise_device="device").eval()
head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
head = head.cuda()
fp = [torch.from_numpy(f).cuda() for f in synth.make_backbone_features(1, B, H, W)]
gt = torch.from_numpy(synth.make_gt_depth(2, B, H, W)).cuda()


def stage_times(n=7):
    names, rows, hrows = None, [], []
    with torch.no_grad():
        for it in range(n + 2):
            ev = [torch.cuda.Event(enable_timing=True)]
            ev[0].record()
            host = [time.perf_counter()]
            def mark():
                e = torch.cuda.Event(enable_timing=True); e.record(); ev.append(e); host.append(time.perf_counter())
            lab = []
            with head._bound.hold():
                g = head.depth_transform.t(gt); mark(); lab.append("encode")
                x = head.aggregate_condition(fp); mark(); lab.append("condition FPN")
                lat = head.pipeline(batch_size=B, device=x.device, dtype=x.dtype, shape=g.shape[-3:], input_args=(x, None, None, None),
                                    num_inference_steps=20, return_dict=False)[0]; mark(); lab.append("20-step loop")
                d = head.depth_transform.inv_t(lat); mark(); lab.append("decode")
                noise = torch.randn(lat.shape, device=lat.device); mark(); lab.append("randn (device)")
                t = torch.randint(0, 1000, (B,), device=lat.device).long(); mark(); lab.append("randint")
                be = head._bound.ensure(lat.device, head.scheduler, need=())
                noisy = head.scheduler.add_noise(lat, noise, t, backend=be); mark(); lab.append("q_sample (dd_add_noise)")
                pred = head.model(noisy, t, x, None, None, None); mark(); lab.append("denoiser call (dd_denoise_once)")
                loss = F.mse_loss(pred, noise); mark(); lab.append("mse_loss")
            torch.cuda.synchronize()
            if it >= 2:
                rows.append([ev[i].elapsed_time(ev[i + 1]) for i in range(len(ev) - 1)])
                hrows.append([(host[i + 1] - host[i]) * 1e3 for i in range(len(host) - 1)])
            names = lab
    med = [sorted(r[i] for r in rows)[len(rows) // 2] for i in range(len(names))]
    hmed = [sorted(r[i] for r in hrows)[len(hrows) // 2] for i in range(len(names))]
    return names, med, hmed


names, med, hmed = stage_times()
print(f"B={B} {prec}, KITTI {H}x{W}: stage medians -- GPU time between hipEvents on the current stream | HOST time the call took to return")
print("  (a host time close to the GPU backlog = a hidden synchronisation in that stage; the host must stay ahead of the GPU)")
for n_, m, hm in zip(names, med, hmed):
    print(f"  {n_:34s} {m:8.3f} ms | host {hm:8.3f} ms")
print(f"  {'sum':34s} {sum(med):8.3f} ms | host {sum(hmed):8.3f} ms")
