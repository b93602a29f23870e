#!/bin/bash
# Round 5, call 13: how large do the stored conv outputs get on the training inputs of seed 320 (f16 storage overflows at 65504)?
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
cat > /tmp/tl9.py <<'PY'
import os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ["DDEPTH_DEVICE_WEIGHTS"] = "1"; os.environ["DDEPTH_STREAMS"] = "1"
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth
variant, prec = "res", "bf16"
chans = (64, 128, 256, 512)
head = dda.DDIMDepthEstimate_Res(precision=prec, inference_steps=20, loss_noise_device="device")
sd = synth.make_state_dict(7240, variant); sd.update(synth.make_fpn_state_dict(7241, in_channels=chans))
head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
head = head.cuda().train()
H, W, B = 352, 1216, 4
fp = [torch.from_numpy(f).cuda() for f in synth.make_backbone_features(7240, B, H, W, in_channels=chans)]
gt = torch.from_numpy(synth.make_gt_depth(7240, B, H, W)).cuda()
with torch.no_grad():
    cond = head.aggregate_condition(fp)          # torch FPN, batch-statistics BatchNorm (train mode)
be = head._bound.ensure(torch.device("cuda", 0), head.scheduler)
sch = dda.DDIMScheduler(); sch.set_timesteps(20)
for seed in (320, 301, 300, 310):
    torch.manual_seed(seed)
    x_T = torch.randn((B, 16, 176, 608), device="cuda")
    for pr in ("bf16", "fp32"):
        be.set_option("hoist_cond", -1)
        tr = be.denoise_trace(x_T, cond, 20, pr)
        worst = {}
        for k in range(20):
            xk = x_T if k == 0 else tr[k - 1]
            t = int(sch.timesteps[k])
            be.set_option("hoist_cond", 0)        # un-hoisted single call: y3 holds the whole conv3 output
            eps = be.denoise_once(xk, torch.full((B,), t, device="cuda", dtype=torch.long), cond, pr)
            for name in ("y1", "y2", "y3", "y4"):
                y = be.debug_fetch(name, B, 176, 608)
                fin = torch.isfinite(y)
                m = float(y[fin].abs().max())
                worst[name] = max(worst.get(name, (0, 0))[0], m), worst.get(name, (0, 0))[1] + int((~fin).sum())
            be.set_option("hoist_cond", -1)
        print(f"[seed {seed} {pr}] finite x_0: {bool(torch.isfinite(tr[-1]).all())} |x_0| max {float(tr[-1][torch.isfinite(tr[-1])].abs().max()):.1f} | max |y| over the 20 steps (non-finite count): " + ", ".join(f"{n} {v[0]:.5g} ({v[1]})" for n, v in worst.items()), flush=True)
PY
timeout 900 python /tmp/tl9.py 2>&1 | grep "^\["
