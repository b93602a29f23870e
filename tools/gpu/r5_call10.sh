#!/bin/bash
# Round 5, call 10: option "check_finite" names the first backward stage whose output is non-finite (16-bit training, fixed parameters)
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
cat > /tmp/tl7.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
variant, prec, streams, N = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
os.environ["DDEPTH_DEVICE_WEIGHTS"] = "1"; os.environ["DDEPTH_STREAMS"] = streams
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
H, W, B = 352, 1216, 4
s0 = 4 if swin else 2
fp = [torch.from_numpy(f).cuda() for f in synth.make_backbone_features(7240, B, H // (s0 // 2), W // (s0 // 2), in_channels=chans)]
gt = torch.from_numpy(synth.make_gt_depth(7240, B, H, W)).cuda()
tag = f"[{variant} {prec} S={streams}]"
for it in range(N):
    torch.manual_seed(300 + it)
    for _, p in named: p.grad = None
    out = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=True)
    head._bound.backend.set_option("check_finite", int(os.environ.get("CHK", "2")))
    which = sys.argv[5] if len(sys.argv) > 5 else "both"
    loss = ((out["pred"] - gt).abs().mean() if which != "ddim" else 0) + (out["ddim_loss"] if which != "depth" else 0)
    try:
        loss.backward()
    except RuntimeError as e:
        print(f"{tag} it {it} ({which}): loss {float(loss):.6g}: {str(e)[:400]}", flush=True)
        break
    bad = [n for n, p in named if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    if bad:
        print(f"{tag} it {it}: non-finite grads WITHOUT a check_finite hit: {bad[:5]}", flush=True); break
else:
    print(f"{tag} {N} iterations ({which}) clean", flush=True)
PY
run() { timeout 900 python /tmp/tl7.py "$@" 2>&1 | grep "^\[" ; }
run res bf16 1 60
run res bf16 1 60
run res bf16 2 60
run swin bf16 1 40
run res f16 1 40
run swin bf16 1 30
