#!/bin/bash
# Round 5, call 6: non-finite gradients after optimizer steps in the 16-bit training step: device vs host weight route, one vs two lanes, which gradients go first
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
cat > /tmp/tl3.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
variant, prec, streams, devw = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
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
fp = [torch.from_numpy(f).cuda().requires_grad_(True) for f in synth.make_backbone_features(7240, B, H // (s0 // 2), W // (s0 // 2), in_channels=chans)]
gt = torch.from_numpy(synth.make_gt_depth(7240, B, H, W)).cuda()
tag = f"[{variant} {prec} S={streams} devw={devw}]"
for it in range(8):
    torch.manual_seed(200 + it)
    opt.zero_grad(set_to_none=True)
    for f in fp: f.grad = None
    pbad = [n for n, p in named if not bool(torch.isfinite(p).all())]
    out = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=True)
    loss = (out["pred"] - gt).abs().mean() + out["ddim_loss"]
    loss.backward()
    bad = [n for n, p in named if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    badfp = [i for i, f in enumerate(fp) if f.grad is not None and not bool(torch.isfinite(f.grad).all())]
    print(f"{tag} it {it}: loss {float(loss):.6g} | non-finite PARAMS before forward {len(pbad)} | non-finite param grads {len(bad)}/{len(named)} {[b for b in bad if b.startswith('model.')][:8]} ... {[b for b in bad if not b.startswith('model.')][:3]} | feature grads {badfp}", flush=True)
    if bad:
        break
    opt.step()
PY
run() { timeout 300 python /tmp/tl3.py "$@" 2>&1 | grep "^\[" ; }
run res bf16 2 1
run res bf16 2 0
run res bf16 1 1
run res f16 2 1
run swin bf16 2 1
run res fp32 2 1
run res bf16 2 1
