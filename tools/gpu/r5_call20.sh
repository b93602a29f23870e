#!/bin/bash
# Round 5, call 20: what the eager training forward costs: train-dp step time with train_graphs = 0 (default) and 1, Swin and Res heads, alternating
cd "$(dirname "$0")/../.."; mkdir -p gpurun_out; export TMPDIR=/tmp
cat > /tmp/tt.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
os.environ["DDEPTH_DEVICE_WEIGHTS"] = "1"; os.environ["DDEPTH_STREAMS"] = "2"
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth
variant, tg = sys.argv[1], int(sys.argv[2])
swin = variant == "swin"
chans = (192, 384, 768, 1536) if swin else (64, 128, 256, 512)
cls = dda.DDIMDepthEstimate_Swin_ADD if swin else dda.DDIMDepthEstimate_Res
head = cls(precision="bf16", inference_steps=20, loss_noise_device="device")
sd = synth.make_state_dict(7240, variant); sd.update(synth.make_fpn_state_dict(7241, in_channels=chans))
head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
head = head.cuda().train()
params = [p for p in head.parameters() if p.requires_grad]
opt = torch.optim.SGD(params, lr=1e-4)
H, W, B = 352, 1216, 4
s0 = 4 if swin else 2
fp = [torch.from_numpy(f).cuda() for f in synth.make_backbone_features(7240, B, H // (s0 // 2), W // (s0 // 2), in_channels=chans)]
gt = torch.from_numpy(synth.make_gt_depth(7240, B, H, W)).cuda()
be = head._bound.ensure(torch.device("cuda", 0), head.scheduler)
be.set_option("train_graphs", tg)
def step():
    opt.zero_grad(set_to_none=True)
    out = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=True)
    loss = (out["pred"] - gt).abs().mean() + out["ddim_loss"]
    loss.backward(); opt.step(); return loss
for _ in range(2): step()
torch.cuda.synchronize(); ts = []
for _ in range(6):
    t0 = time.perf_counter(); l = step(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"[{variant} train_graphs={tg}] step ms: median {sorted(ts)[3]:.2f} (min {min(ts):.2f} max {max(ts):.2f}); loss finite {bool(torch.isfinite(l))}; graph launches {be.counter('graph_launches')}", flush=True)
PY
for i in 1 2; do for v in swin res; do for tg in 0 1; do timeout 300 python /tmp/tt.py $v $tg 2>&1 | grep "^\["; done; done; done
