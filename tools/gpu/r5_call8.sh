#!/bin/bash
# Round 5, call 8: first anomaly of the 16-bit training loop, many iterations, by configuration
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
cat > /tmp/tl5.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
variant, prec, streams, devw, N = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
os.environ["DDEPTH_DEVICE_WEIGHTS"] = devw; os.environ["DDEPTH_STREAMS"] = streams
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
agg = head.aggregate_condition
def agg2(fp_, neck=False):
    c = agg(fp_, neck); cap["cond_finite"] = bool(torch.isfinite(c).all()); cap["cond_max"] = float(c.detach().abs().max()); return c
head.aggregate_condition = agg2
tag = f"[{variant} {prec} S={streams} devw={devw}]"
grp = lambda n: "model" if n.startswith("model.") else "codec" if n.startswith("depth_transform.") else "fpn"
for it in range(N):
    torch.manual_seed(300 + it)
    opt.zero_grad(set_to_none=True)
    pbad = sorted({grp(n) for n, p in named if not bool(torch.isfinite(p).all())})
    out = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=True)
    l1, l2 = (out["pred"] - gt).abs().mean(), out["ddim_loss"]
    loss = l1 + l2
    anomalies = []
    if pbad: anomalies.append(f"non-finite PARAMS {pbad}")
    if not cap["cond_finite"]: anomalies.append("non-finite cond from the torch FPN")
    if not bool(torch.isfinite(out["pred"]).all()): anomalies.append(f"non-finite pred (images {[i for i in range(B) if not bool(torch.isfinite(out['pred'][i]).all())]})")
    if not bool(torch.isfinite(l2)): anomalies.append("non-finite ddim_loss")
    if bool(torch.isfinite(loss)):
        loss.backward()
        gb = {}
        for n, p in named:
            if p.grad is not None and not bool(torch.isfinite(p.grad).all()): gb.setdefault(grp(n), []).append(n)
        if gb: anomalies.append("non-finite GRADS " + str({k: (len(v), v[:3]) for k, v in gb.items()}))
        gn = sum(float(p.grad.float().norm() ** 2) for _, p in named if p.grad is not None) ** 0.5
    else:
        gn = float("nan")
    if anomalies or it % 10 == 0:
        print(f"{tag} it {it}: loss {float(loss):.6g} grad norm {gn:.4g} cond max {cap['cond_max']:.4g} | " + ("; ".join(anomalies) if anomalies else "ok"), flush=True)
    if anomalies:
        break
    opt.step()
else:
    print(f"{tag} {N} iterations without an anomaly", flush=True)
PY
run() { timeout 600 python /tmp/tl5.py "$@" 2>&1 | grep "^\[" ; }
run swin bf16 2 1 40
run swin bf16 1 1 40
run swin bf16 2 0 40
run swin fp32 2 1 25
run res bf16 2 1 40
run swin bf16 2 1 40
