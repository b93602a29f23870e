#!/usr/bin/env python3
"""Round 6: what does the round-5 training harness need, beside the replayed hipGraph of the trajectory-keeping forward, for its intermittent non-finite
gradients (profiles/r05_experiments.md section 4)?  The torch-free C++ driver (tools/micro/train_graph_repro.cpp) never fails; the full head does.
One process = 40 iterations of one ARM, KITTI size, B = 4, bf16, option train_graphs from argv:

  lib       the library alone, driven from Python on torch-allocated tensors: dd_denoise(keep) -> dd_denoise_once(keep) -> [sync: "loss"] ->
            dd_denoise_once_backward -> dd_denoise_backward -> [sync: gradients]; no torch kernel inside an iteration except the finiteness reductions
  autograd  the same two calls as modules.CNNDDIMPipiline / ScheduledCNNRefine in .train() (autograd Functions: the backward runs on torch's autograd
            thread), the condition map a leaf that requires grad; no FPN, no codec
  head      the round-5 harness: the whole DDIMDepthEstimate_Res training step (FPN + codec in PyTorch with batch-statistics BatchNorm = MIOpen / ATen kernels
            between the library calls)
  head_nomiopen   the same with torch.backends.cudnn.enabled = False (ATen's own convolution / batch-norm kernels instead of MIOpen's)
  head_bn_native  MIOpen for the convolutions only: every BatchNorm2d forward runs under cudnn.flags(enabled=False) (ATen's batch-norm kernels)
  head_conv_native   MIOpen for the batch norms only: every Conv2d / ConvTranspose2d forward runs under cudnn.flags(enabled=False)
  head_eval_bn    the head in .train() but its BatchNorm layers in .eval() (running statistics: no batch-norm TRAINING kernels; convolutions via MIOpen)

    python tools/nan_arms.py <arm> <train_graphs 0|1> [iterations 40]
prints one line: "[arm tg=N] <iterations> iterations, bad iterations: K (first at I)"."""
import os, sys
os.environ.setdefault("DDEPTH_STREAMS", "1"); os.environ.setdefault("DDEPTH_DEVICE_WEIGHTS", "1"); os.environ["DDEPTH_GRAD_GUARD"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
import diffusiondepth_amd as dda
from diffusiondepth_amd import synth, modules as M

arm, tg, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]) if len(sys.argv) > 3 else 40
B, H, W, T = 4, 352, 1216, 20
h, w = synth.latent_hw(H, W)
dev = torch.device("cuda", 0)
bad = []
finite = lambda ts: all(bool(torch.isfinite(t).all()) for t in ts if t is not None)

if arm == "lib":
    be = dda.HipDenoiser(dev); be.load_state_dict(synth.make_state_dict(7240)); be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
    be.set_option("train_graphs", tg)
    inp = synth.make_inputs(7240, B, h, w)
    x_T, cond = torch.from_numpy(inp["x_T"]).to(dev), torch.from_numpy(inp["cond"]).to(dev)
    g0 = torch.randn_like(x_T) * 1e-3
    tt = torch.tensor([37, 248, 459, 670], device=dev)
    for it in range(N):
        x0 = be.denoise(x_T, cond, T, "bf16", keep_trajectory=True); tk = be.last_trajectory_ticket
        eps = be.denoise_once(x0, tt, cond, "bf16", keep_trajectory=True); tk1 = be.last_trajectory_ticket
        ok = finite([x0, eps])
        be.zero_grad()
        g1 = be.denoise_once_backward(x0, tt, cond, g0, "bf16", need_grad_x=True, need_grad_cond=True, trajectory_ticket=tk1)
        p1 = list(be.grads().values())
        be.zero_grad()
        g2 = be.denoise_backward(x_T, cond, g0, T, "bf16", need_grad_xT=False, need_grad_cond=True, trajectory_ticket=tk)
        p2 = list(be.grads().values())
        if not (ok and finite(list(g1) + list(g2) + p1 + p2)):
            bad.append(it)
elif arm == "autograd":
    bound = M.HipBound("res")
    model = M.ScheduledCNNRefine(bound=bound, precision="bf16").to(dev).train()
    sd = synth.make_state_dict(7240)
    model.load_state_dict({k[len("model."):]: torch.from_numpy(v) for k, v in sd.items() if k.startswith("model.")})
    sched = dda.DDIMScheduler(num_train_timesteps=1000, clip_sample=False)
    pipe = M.CNNDDIMPipiline(model, sched)
    bound.ensure(dev, sched).set_option("train_graphs", tg)
    inp = synth.make_inputs(7240, B, h, w)
    x_T = torch.from_numpy(inp["x_T"]).to(dev)
    cond0 = torch.from_numpy(inp["cond"]).to(dev)
    tt = torch.tensor([37, 248, 459, 670], device=dev)
    params = [p for p in model.parameters() if p.requires_grad]
    for it in range(N):
        for p in params:
            p.grad = None
        cond = cond0.clone().requires_grad_(True)
        out, = pipe(batch_size=B, device=dev, dtype=torch.float32, shape=(16, h, w), input_args=(cond, None, None, None), num_inference_steps=T, return_dict=False, x_T=x_T)
        eps = model(out.detach(), tt, cond, None, None, None)
        loss = out.abs().mean() * 1e-3 + eps.mean()
        if not bool(torch.isfinite(loss)):
            bad.append(it); continue
        loss.backward()
        if not finite([cond.grad] + [p.grad for p in params]):
            bad.append(it)
else:
    if arm == "head_nomiopen":
        torch.backends.cudnn.enabled = False

    def native(cls):
        f = cls.forward
        def fwd(self, *a, **k):
            with torch.backends.cudnn.flags(enabled=False):
                return f(self, *a, **k)
        cls.forward = fwd
    if arm == "head_bn_native":
        native(torch.nn.BatchNorm2d)
    if arm == "head_conv_native":
        native(torch.nn.Conv2d); native(torch.nn.ConvTranspose2d)
    head = dda.DDIMDepthEstimate_Res(precision="bf16", inference_steps=T, loss_noise_device="device")
    sd = synth.make_state_dict(7240); sd.update(synth.make_fpn_state_dict(7241))
    head.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    head = head.to(dev).train()
    if arm == "head_eval_bn":
        for m in head.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.eval()
    named = [(n, p) for n, p in head.named_parameters() if p.requires_grad]
    fp = [torch.from_numpy(f).to(dev) for f in synth.make_backbone_features(7240, B, H, W)]
    gt = torch.from_numpy(synth.make_gt_depth(7240, B, H, W)).to(dev)
    head._bound.ensure(dev, head.scheduler).set_option("train_graphs", tg)
    if os.environ.get("GRAPH_FENCE"):        # library option graph_fence: 1 = synchronise in front of the graph launch, 2 = behind it, 4 = launch on the handle's own stream
        head._bound.backend.set_option("graph_fence", int(os.environ["GRAPH_FENCE"])); arm += " fence=" + os.environ["GRAPH_FENCE"]
    for it in range(N):
        torch.manual_seed(320)
        for _, p in named:
            p.grad = None
        out = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=True)
        loss = (out["pred"] - gt).abs().mean() + out["ddim_loss"]
        if not bool(torch.isfinite(loss)):
            bad.append(it); continue
        loss.backward()
        if not finite([p.grad for _, p in named]):
            bad.append(it)
ctr = ""
try:
    b_ = be if arm == "lib" else (bound.backend if arm == "autograd" else head._bound.backend)
    ctr = "; " + ", ".join(f"{k} {b_.counter(k)}" for k in ("graph_launches", "eager_loops", "trajectory_reuses", "plans"))
except Exception:  # noqa: BLE001
    pass
envs = " ".join(f"{k}={os.environ[k]}" for k in ("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "HIP_FORCE_DEV_KERNARG", "GPU_MAX_HW_QUEUES", "AMD_DIRECT_DISPATCH") if k in os.environ)
print(f"[{arm} tg={tg}{' ' + envs if envs else ''}] {N} iterations, bad iterations: {len(bad)}" + (f" (first at {bad[0]}, last at {bad[-1]})" if bad else "") + ctr, flush=True)
