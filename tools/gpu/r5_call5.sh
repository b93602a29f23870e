#!/bin/bash
# Round 5, call 5: where do the non-finite gradients of the 16-bit training step come from?  fixed parameters (no optimizer step), seeds x lanes x precision
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
cat > /tmp/tl2.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
streams = sys.argv[3]
os.environ.setdefault("DDEPTH_DEVICE_WEIGHTS", "1"); os.environ["DDEPTH_STREAMS"] = streams
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth
variant, prec = sys.argv[1], sys.argv[2]
swin = variant == "swin"
chans = (192, 384, 768, 1536) if swin else (64, 128, 256, 512)
cls = dda.DDIMDepthEstimate_Swin_ADD if swin else dda.DDIMDepthEstimate_Res
head = cls(precision=prec, inference_steps=20, loss_noise_device="device")
sd = synth.make_state_dict(7240, variant); sd.update(synth.make_fpn_state_dict(7241, in_channels=chans))
head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
head = head.cuda().train()
named = [(n, p) for n, p in head.named_parameters() if p.requires_grad]
H, W, B = 352, 1216, int(sys.argv[4]) if len(sys.argv) > 4 else 4
s0 = 4 if swin else 2
fp = [torch.from_numpy(f).cuda().requires_grad_(True) for f in synth.make_backbone_features(7240, B, H // (s0 // 2), W // (s0 // 2), in_channels=chans)]
gt = torch.from_numpy(synth.make_gt_depth(7240, B, H, W)).cuda()
for it in range(5):
    torch.manual_seed(100 + it)
    for _, p in named: p.grad = None
    for f in fp: f.grad = None
    out = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=True)
    which = sys.argv[5] if len(sys.argv) > 5 else "both"
    loss = ((out["pred"] - gt).abs().mean() if which != "ddim" else 0) + (out["ddim_loss"] if which != "depth" else 0)
    loss.backward()
    bad = [n for n, p in named if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    badfp = [i for i, f in enumerate(fp) if f.grad is not None and not bool(torch.isfinite(f.grad).all())]
    mx = max(float(p.grad.abs().max()) for _, p in named if p.grad is not None and bool(torch.isfinite(p.grad).all())) if len(bad) < len(named) else float("nan")
    print(f"[{variant} {prec} S={streams} B={B} {which}] seed {100 + it}: loss {float(loss):.5g} | non-finite param grads {len(bad)}/{len(named)} {bad[:6]} | non-finite feature grads {badfp} | max finite |grad| {mx:.3g}", flush=True)
PY
run() { timeout 300 python /tmp/tl2.py "$@" 2>&1 | grep "^\[" ; }
run res bf16 2
run res bf16 1
run res f16 2
run res bf16 2 4 depth
run res bf16 2 4 ddim
run res bf16 2 1
run swin bf16 2
