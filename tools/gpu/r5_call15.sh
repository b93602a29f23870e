#!/bin/bash
# Round 5, call 15: what happens around the failing iterations (20..34 with a fixed seed): per-iteration finiteness of every stage of the head + library counters
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
cat > /tmp/tl10.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ["DDEPTH_DEVICE_WEIGHTS"] = "1"; os.environ["DDEPTH_STREAMS"] = "1"
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth
prec, mode = sys.argv[1], sys.argv[2]
chans = (64, 128, 256, 512)
head = dda.DDIMDepthEstimate_Res(precision=prec, inference_steps=20, loss_noise_device="device")
sd = synth.make_state_dict(7240, "res"); sd.update(synth.make_fpn_state_dict(7241, in_channels=chans))
head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
head = head.cuda().train()
named = [(n, p) for n, p in head.named_parameters() if p.requires_grad]
H, W, B = 352, 1216, 4
fp = [torch.from_numpy(f).cuda() for f in synth.make_backbone_features(7240, B, H, W, in_channels=chans)]
gt = torch.from_numpy(synth.make_gt_depth(7240, B, H, W)).cuda()
be = head._bound.ensure(torch.device("cuda", 0), head.scheduler)
cap = {}
orig = head.pipeline.__class__.__call__
def wrapped(self, *a, **kw):
    r = orig(self, *a, **kw); cap["x0_finite"] = bool(torch.isfinite(r[0]).all()); cap["x0_max"] = float(r[0].detach().abs().max()); return r
head.pipeline.__class__.__call__ = wrapped
import torch.nn.functional as F
for it in range(40):
    torch.manual_seed(320)
    for _, p in named: p.grad = None
    mem = torch.cuda.memory_allocated() >> 20
    free, total = torch.cuda.mem_get_info()
    out = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=True)
    l1, l2 = (out["pred"] - gt).abs().mean(), out["ddim_loss"]
    ok = bool(torch.isfinite(l1 + l2))
    msg = f"it {it}: x0 finite {cap['x0_finite']} max {cap['x0_max']:.4g} | pred finite {bool(torch.isfinite(out['pred']).all())} | ddim {float(l2):.5g} | torch MiB {mem} free GiB {free / 2**30:.1f} | plans {be.counter('plans')} graphs {be.counter('graph_launches')} eager {be.counter('eager_loops')} capfail {be.counter('graph_capture_failures')} reuses {be.counter('trajectory_reuses')} ticket {be.counter('trajectory_ticket')}"
    if mode == "bwd" and ok:
        (l1 + l2).backward()
        bad = [n for n, p in named if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
        msg += f" | non-finite grads {len(bad)}"
    if it >= 16 or not ok:
        print(f"[{prec} {mode}] " + msg, flush=True)
PY
timeout 900 python /tmp/tl10.py bf16 bwd 2>&1 | grep "^\["
timeout 900 python /tmp/tl10.py bf16 fwd 2>&1 | grep "^\[" | tail -n 8
