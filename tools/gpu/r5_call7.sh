#!/bin/bash
# Round 5, call 7: localise the non-finite value of the Swin bf16 training forward (seed 201, after one SGD step)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
cat > /tmp/tl4.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
variant, prec = sys.argv[1], sys.argv[2]
os.environ["DDEPTH_DEVICE_WEIGHTS"] = "1"; os.environ["DDEPTH_STREAMS"] = "2"
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth
swin = variant == "swin"
chans = (192, 384, 768, 1536) if swin else (64, 128, 256, 512)
cls = dda.DDIMDepthEstimate_Swin_ADD if swin else dda.DDIMDepthEstimate_Res
head = cls(precision=prec, inference_steps=20, loss_noise_device="device")
sd = synth.make_state_dict(7240, variant); sd.update(synth.make_fpn_state_dict(7241, in_channels=chans))
head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
head = head.cuda().train()
named = [(n, p) for n, p in head.named_parameters() if p.requires_grad]
opt = torch.optim.SGD([p for _, p in named], lr=1e-4)
H, W, B = 352, 1216, 4
s0 = 4 if swin else 2
fp = [torch.from_numpy(f).cuda() for f in synth.make_backbone_features(7240, B, H // (s0 // 2), W // (s0 // 2), in_channels=chans)]
gt = torch.from_numpy(synth.make_gt_depth(7240, B, H, W)).cuda()
cap = {}
orig = head.pipeline.__class__.__call__
def wrapped(self, batch_size, device, dtype, shape, input_args, **kw):
    x_T = torch.randn((batch_size, *shape), device=device, dtype=dtype)
    cap["x_T"], cap["cond"] = x_T.detach().clone(), input_args[0].detach().clone()
    return orig(self, batch_size, device, dtype, shape, input_args, x_T=x_T, **kw)
head.pipeline.__class__.__call__ = wrapped
be = None
for it in range(12):
    torch.manual_seed(200 + it)
    opt.zero_grad(set_to_none=True)
    out = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=True)
    l1, l2 = (out["pred"] - gt).abs().mean(), out["ddim_loss"]
    print(f"it {it}: depth L1 {float(l1):.6g} ddim {float(l2):.6g} | cond max {float(cap['cond'].abs().max()):.4g} finite cond {bool(torch.isfinite(cap['cond']).all())} | pred finite {bool(torch.isfinite(out['pred']).all())}", flush=True)
    if not torch.isfinite(l1 + l2):
        break
    (l1 + l2).backward()
    opt.step()
be = head._bound.backend
x_T, cond = cap["x_T"], cap["cond"]
print("analysing the failing forward: which images are non-finite:", [bool(torch.isnan(out["pred"][i]).any()) for i in range(B)], flush=True)
be.set_option("hoist_cond", 0)      # the reference's order, as the training plans run it
for pr in (prec, "fp32"):
    tr = be.denoise_trace(x_T, cond, 20, pr)
    bad = [(k, [bool(torch.isnan(tr[k][i]).any()) for i in range(B)]) for k in range(20)]
    first = next((k for k, b in bad if any(b)), None)
    print(f"[{pr}] trace (eager, reference order): first step with a NaN state: {first}; |x| max per step:", [round(float(tr[k][torch.isfinite(tr[k])].abs().max()), 1) for k in range(0, 20, 3)], flush=True)
    if first is not None:
        xk = x_T if first == 0 else tr[first - 1]
        ts = dda.DDIMScheduler(); ts.set_timesteps(20)
        t = int(ts.timesteps[first])
        eps = be.denoise_once(xk, torch.full((B,), t, device="cuda", dtype=torch.long), cond, pr)
        print(f"[{pr}] step {first} (t = {t}): eps non-finite per image {[bool(torch.isnan(eps[i]).any()) for i in range(B)]}; input |x| max {float(xk.abs().max()):.4g}", flush=True)
        for name, C in (("y1", 64), ("y2", 256), ("y3", 64), ("y4", 16)):
            try:
                y = be.debug_fetch(name, B, x_T.shape[2], x_T.shape[3])
                fin = torch.isfinite(y)
                print(f"    {name}: non-finite values {int((~fin).sum())} (per image {[int((~fin[i]).sum()) for i in range(B)]}), max finite |.| {float(y[fin].abs().max()):.5g}", flush=True)
            except Exception as e:
                print("    ", name, "fetch failed:", e)
PY
timeout 600 python /tmp/tl4.py swin bf16 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -n 30
timeout 600 python /tmp/tl4.py swin f16 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -n 30
