"""Round 6: N training iterations of the KITTI-size head (forward, backward, no host synchronisation inside an iteration; tests/test_zzzz_gpu_soak.py's loop with N from argv):
finiteness of the loss and of every gradient per iteration.  argv: variant res|swin, precision, N"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_zzzz_gpu_soak as S
variant, prec, N = sys.argv[1], sys.argv[2], int(sys.argv[3])
head, fp, gt = S._kitti_head(variant, precision=prec, loss_noise_device="device")
head = head.train()
named = [(n, p) for n, p in head.named_parameters() if p.requires_grad]
flags = torch.ones((N, 2), device="cuda")
for it in range(N):
    torch.manual_seed(320)
    for _, p in named:
        p.grad = None
    out = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=True)
    loss = (out["pred"] - gt).abs().mean() + out["ddim_loss"]
    flags[it, 0] = torch.isfinite(loss).float()
    loss.backward()
    for _, p in named:
        if p.grad is not None:
            flags[it, 1] *= torch.isfinite(p.grad).all().float()
f = flags.cpu().numpy()
bad = [i for i in range(N) if f[i].min() == 0.0]
be = head._bound.backend
print(f"train {variant} {prec}: {len(bad)} of {N} iterations non-finite {bad[:5]}; graph_launches {be.counter('graph_launches')} eager_loops {be.counter('eager_loops')} lane_overlap {be.counter('lane_overlap')}", flush=True)
