#!/bin/bash
# Round 5, call 23: is it the memset NODE of the captured loop?  training forward as a graph whose GroupNorm sums are zeroed by a kernel node vs by the memset node, 8 processes each
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
cat > /tmp/tl8.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
variant, prec, streams, N, opts = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
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
FIX = int(os.environ.get('FIXSEED', '-1'))
tag = f"[seed{os.environ.get('FIXSEED','')} {variant} {prec} S={streams} {opts} {os.environ.get('AMD_SERIALIZE_KERNEL','')}{os.environ.get('HIP_LAUNCH_BLOCKING','')}]"
be = head._bound.ensure(torch.device("cuda", 0), head.scheduler)
for kv in filter(None, opts.split(",")):
    if kv == "none": continue
    k, v = kv.split("="); be.set_option(k, int(v))
side = torch.cuda.Stream() if os.environ.get('SIDE') else None
ctx = torch.cuda.stream(side) if side is not None else __import__('contextlib').nullcontext()
ctx.__enter__()
if side is not None: tag = '[SIDE-STREAM ' + tag[1:]
for it in range(N):
    torch.manual_seed(300 + it if FIX < 0 else FIX)
    for _, p in named: p.grad = None
    out = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=True)
    loss = (out["pred"] - gt).abs().mean() + out["ddim_loss"]
    if not bool(torch.isfinite(loss)):
        print(f"{tag} it {it}: non-finite LOSS", flush=True); nfail = globals().get('nfail', 0) + 1; globals()['nfail'] = nfail; continue
    loss.backward()
    bad = [n for n, p in named if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    if bad:
        print(f"{tag} it {it}: non-finite grads: {len(bad)}/{len(named)} last-model {[b for b in bad if b.startswith('model.')][-2:]}", flush=True); globals()['nfail'] = globals().get('nfail', 0) + 1
print(f"{tag} {N} iterations, failures: {globals().get('nfail', 0)}", flush=True)
PY

run() { timeout 900 python /tmp/tl8.py "$@" 2>&1 | grep "^\[" | grep -v "non-finite LOSS\|non-finite grads" | tail -n 1; }
for i in 1 2 3 4 5 6 7 8; do
FIXSEED=320 run res bf16 1 40 train_graphs=1,graph_memset_kernel=1
FIXSEED=320 run res bf16 1 40 train_graphs=1
done
