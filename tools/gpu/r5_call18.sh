#!/bin/bash
# Round 5, call 18: per-iteration finiteness of every stage recorded ON THE DEVICE (no host synchronisation inside the loop), read at the end
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
cat > /tmp/tl11.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ["DDEPTH_DEVICE_WEIGHTS"] = "1"; os.environ["DDEPTH_STREAMS"] = sys.argv[1]
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth, head as headmod
import torch.nn.functional as F
chans = (64, 128, 256, 512)
head = dda.DDIMDepthEstimate_Res(precision="bf16", inference_steps=20, loss_noise_device=sys.argv[2])
sd = synth.make_state_dict(7240, "res"); sd.update(synth.make_fpn_state_dict(7241, in_channels=chans))
head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
head = head.cuda().train()
named = [(n, p) for n, p in head.named_parameters() if p.requires_grad]
H, W, B, N = 352, 1216, 4, 40
fp = [torch.from_numpy(f).cuda() for f in synth.make_backbone_features(7240, B, H, W, in_channels=chans)]
gt = torch.from_numpy(synth.make_gt_depth(7240, B, H, W)).cuda()
NAMES = ["cond", "x0(loop)", "pred(decoder)", "noisy(q_sample)", "eps(single call)", "noise", "ddim_loss", "grads model", "grads fpn", "grads codec"]
flags = torch.ones((N, len(NAMES)), device="cuda")
cur = {"it": 0}
fin = lambda t: torch.isfinite(t).all().float()
agg = head.aggregate_condition
def agg2(fp_, neck=False):
    c = agg(fp_, neck); flags[cur["it"], 0] = fin(c); return c
head.aggregate_condition = agg2
orig = head.pipeline.__class__.__call__
def wrapped(self, *a, **kw):
    r = orig(self, *a, **kw); flags[cur["it"], 1] = fin(r[0]); return r
head.pipeline.__class__.__call__ = wrapped
add_noise = head.scheduler.add_noise
def add_noise2(x0, noise, t, backend=None):
    flags[cur["it"], 5] = fin(noise); r = add_noise(x0, noise, t, backend=backend); flags[cur["it"], 3] = fin(r); return r
head.scheduler.add_noise = add_noise2
mfwd = head.model.forward
def mfwd2(*a, **kw):
    r = mfwd(*a, **kw); flags[cur["it"], 4] = fin(r); return r
head.model.forward = mfwd2
grp = lambda n: 7 if n.startswith("model.") else 9 if n.startswith("depth_transform.") else 8
for it in range(N):
    cur["it"] = it
    torch.manual_seed(320)
    for _, p in named: p.grad = None
    out = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=True)
    flags[it, 2] = fin(out["pred"]); flags[it, 6] = fin(out["ddim_loss"])
    loss = (out["pred"] - gt).abs().mean() + out["ddim_loss"]
    loss.backward()
    for n, p in named:
        if p.grad is not None: flags[it, grp(n)] *= fin(p.grad)
torch.cuda.synchronize()
f = flags.cpu()
bad_its = [i for i in range(N) if float(f[i].min()) == 0.0]
print(f"[S={sys.argv[1]} noise={sys.argv[2]}] iterations with a non-finite stage: {bad_its}", flush=True)
for i in bad_its[:4] + bad_its[-2:]:
    print(f"    it {i}: non-finite: {[NAMES[j] for j in range(len(NAMES)) if float(f[i, j]) == 0.0]}", flush=True)
PY
for i in 1 2 3 4 5; do timeout 600 python /tmp/tl11.py 1 device 2>&1 | grep "^\[\|^    "; done
for i in 1 2; do timeout 600 python /tmp/tl11.py 1 cpu 2>&1 | grep "^\[\|^    "; done
