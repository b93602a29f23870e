#!/bin/bash
# Round 5, call 4: the Swin train-dp line's non-finite loss -- this round's library vs round 4's (build_variants/libddepth_base.so), loss per step
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
cat > /tmp/tl.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth
os.environ.setdefault("DDEPTH_DEVICE_WEIGHTS", "1"); os.environ["DDEPTH_STREAMS"] = "2"
variant = sys.argv[1]; prec = sys.argv[2]
swin = variant == "swin"
chans = (192, 384, 768, 1536) if swin else (64, 128, 256, 512)
cls = dda.DDIMDepthEstimate_Swin_ADD if swin else dda.DDIMDepthEstimate_Res
head = cls(precision=prec, inference_steps=20, loss_noise_device="device")
sd = synth.make_state_dict(7240, variant); sd.update(synth.make_fpn_state_dict(7241, in_channels=chans))
head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
head = head.cuda().train()
params = [p for p in head.parameters() if p.requires_grad]
opt = torch.optim.SGD(params, lr=1e-4)
H, W, B = 352, 1216, 4
s0 = 4 if swin else 2
fp = [torch.from_numpy(f).cuda() for f in synth.make_backbone_features(7240, B, H // (s0 // 2), W // (s0 // 2), in_channels=chans)]
gt = torch.from_numpy(synth.make_gt_depth(7240, B, H, W)).cuda()
for it in range(6):
    opt.zero_grad(set_to_none=True)
    out = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=True)
    l1 = (out["pred"] - gt).abs().mean(); l2 = out["ddim_loss"]
    loss = l1 + l2
    loss.backward()
    gn = sum(float(p.grad.float().norm() ** 2) for p in params if p.grad is not None) ** 0.5
    nbad = sum(int((~torch.isfinite(p.grad)).sum()) for p in params if p.grad is not None)
    print(f"[{os.path.basename(os.environ.get('DDEPTH_LIBRARY','default'))} {variant} {prec}] it {it}: depth L1 {float(l1):.6g} ddim {float(l2):.6g} pred max {float(out['pred'].max()):.4g} finite pred {bool(torch.isfinite(out['pred']).all())} grad norm {gn:.4g} non-finite grads {nbad}", flush=True)
    opt.step()
PY
for lib in diffusiondepth_amd/libddepth_hip.so build_variants/libddepth_base.so; do
  for cfg in "swin bf16" "swin fp32" "res bf16"; do
    DDEPTH_LIBRARY=$lib timeout 300 python /tmp/tl.py $cfg 2>&1 | grep -v amdgpu.ids | tail -n 8
  done
done
