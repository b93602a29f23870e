"""torch-CPU port of the hot path  --  TEST / BASELINE INFRASTRUCTURE ONLY (never imported by the product).

Same algorithm as oracle/ddim_oracle.py, restated with the stock torch.nn.functional ops the reference
itself executes on CPU (F.conv2d, F.group_norm, F.conv_transpose2d, F.batch_norm), fp32, all host cores.
Used (a) as bench.py's ``cpu_baseline`` (kind "port": the reference tree is not present on the GPU box,
so its own classes cannot be timed there; this port issues the identical sequence of torch kernels), and
(b) as the full-size (NYU / KITTI) checker where the fp64 NumPy oracle would take minutes.
Pinned to the golden vectors by tests/test_oracle_golden.py::test_torch_port_*.

Reference lines restated: src/model/head/ddim_depth_estimate_res.py:248-297 (loop), :300-344 (denoiser);
src/model/diffusers/schedulers/scheduling_ddim.py:215-229, 285-326, 355-376; src/model/ops/depth_transform.py:10-35.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def to_torch_sd(sd, dtype=torch.float32):
    return {k: torch.as_tensor(np.asarray(v)).to(dtype) for k, v in sd.items()}


def make_alphas_cumprod(n=1000, beta_start=1e-4, beta_end=0.02):
    betas = torch.linspace(beta_start, beta_end, n, dtype=torch.float32)        # scheduling_ddim.py:131
    return torch.cumprod(1.0 - betas, dim=0)                                    # :143-144


def timesteps(T, n=1000):
    return (np.arange(0, T) * (n // T)).round()[::-1].copy().astype(np.int64)   # :215-229


def denoiser(sd, x, t, cond, variant="res"):
    """ScheduledCNNRefine.forward (…res.py:324-344; variant="swin": …swin_addHAHI.py:364-382 with the
    UpSample_add fuse convB(convA(bilinear_up(feat) + NE(x)))); t: 0-d / int or (B,) tensor."""
    emb = F.embedding(torch.as_tensor(t, dtype=torch.long), sd["model.time_embedding.weight"])[..., None, None]
    feat = cond + emb
    y = F.conv2d(x, sd["model.noise_embedding.0.weight"], sd["model.noise_embedding.0.bias"], padding=1)
    y = F.relu(F.group_norm(y, 4, sd["model.noise_embedding.1.weight"], sd["model.noise_embedding.1.bias"]))
    y = F.conv2d(y, sd["model.noise_embedding.3.weight"], sd["model.noise_embedding.3.bias"], padding=1)
    y = F.relu(F.group_norm(y, 4, sd["model.noise_embedding.4.weight"], sd["model.noise_embedding.4.bias"]))
    if variant == "res":
        f = feat + y
    else:
        up = F.interpolate(feat, size=[y.size(2), y.size(3)], mode="bilinear", align_corners=True)
        f = F.conv2d(up + y, sd["model.upsample_fuse.convA.conv.weight"], sd["model.upsample_fuse.convA.conv.bias"], padding=1)
        f = F.conv2d(f, sd["model.upsample_fuse.convB.conv.weight"], sd["model.upsample_fuse.convB.conv.bias"], padding=1)
    y = F.conv2d(f, sd["model.pred.0.weight"], sd["model.pred.0.bias"], padding=1)
    y = F.relu(F.group_norm(y, 4, sd["model.pred.1.weight"], sd["model.pred.1.bias"]))
    y = F.conv2d(y, sd["model.pred.3.weight"], sd["model.pred.3.bias"], padding=1)
    return F.relu(F.group_norm(y, 4, sd["model.pred.4.weight"], sd["model.pred.4.bias"]))


def ddim_step(acp, eps, t, x, ratio):
    """DDIMScheduler.step, eta=0, use_clipped_model_output=True, literal formula order (:285-326)."""
    prev = t - ratio
    a_t = acp[t]
    a_prev = acp[prev] if prev >= 0 else torch.tensor(1.0)
    beta_t = 1 - a_t
    x0 = (x - beta_t ** 0.5 * eps) / a_t ** 0.5
    eps2 = (x - a_t ** 0.5 * x0) / beta_t ** 0.5
    return a_prev ** 0.5 * x0 + (1 - a_prev) ** 0.5 * eps2


@torch.no_grad()
def ddim_loop(sd, x_T, cond, T=20, n_train=1000, variant="res"):
    acp = make_alphas_cumprod(n_train)
    x = torch.as_tensor(x_T)
    cond = torch.as_tensor(cond)
    for t in timesteps(T, n_train):
        eps = denoiser(sd, x, int(t), cond, variant)
        x = ddim_step(acp, eps, int(t), x, n_train // T)
    return x


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


@torch.no_grad()
def encode(sd, depth):
    p = "depth_transform.conv_transform."
    y = F.conv2d(torch.as_tensor(depth), sd[p + "0.0.weight"], None, stride=2, padding=1)
    y = F.leaky_relu(_bn(sd, p + "0.1", y), 0.2)
    y = _bn(sd, p + "1.1", F.conv2d(y, sd[p + "1.0.weight"], None, padding=1))
    return torch.tanh(y)


@torch.no_grad()
def decode(sd, latent, eps=1e-6):
    p = "depth_transform.conv_inv_transform."
    y = F.conv_transpose2d(torch.as_tensor(latent), sd[p + "0.weight"], sd[p + "0.bias"], stride=2, padding=1)
    y = F.relu(_bn(sd, p + "1", y))
    y = torch.sigmoid(F.conv2d(y, sd[p + "3.0.weight"], sd[p + "3.0.bias"], padding=1))
    return 1.0 / y.clamp(eps) - 1


def denoiser_vjp(sd, x, t, cond, grad_eps, variant="res"):
    """Vector-Jacobian product of `denoiser` by torch autograd (what loss.backward() sends through one call of the
    reference's ScheduledCNNRefine, ...res.py:211 / src/main.py:232-241).  Returns (eps, grad_x, grad_cond, {name: grad})."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("model.")}
    full = dict(sd)
    full.update(params)
    x = torch.as_tensor(x).clone().requires_grad_(True)
    cond = torch.as_tensor(cond).clone().requires_grad_(True)
    eps = denoiser(full, x, t, cond, variant)
    eps.backward(torch.as_tensor(grad_eps))
    return eps.detach(), x.grad, cond.grad, {k: v.grad for k, v in params.items() if v.grad is not None}


def ddim_loop_vjp(sd, x_T, cond, grad_x0, T=20, n_train=1000, variant="res"):
    """Autograd through the whole T-step loop (the reference trains through it: the pipeline output is not detached,
    ...res.py:124-169).  Returns (x_0, grad_xT, grad_cond, {name: grad})."""
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k.startswith("model.")}
    full = dict(sd)
    full.update(params)
    acp = make_alphas_cumprod(n_train)
    x = torch.as_tensor(x_T).clone().requires_grad_(True)
    cond = torch.as_tensor(cond).clone().requires_grad_(True)
    cur = x
    for t in timesteps(T, n_train):
        eps = denoiser(full, cur, int(t), cond, variant)
        cur = ddim_step(acp, eps, int(t), cur, n_train // T)
    cur.backward(torch.as_tensor(grad_x0))
    return cur.detach(), x.grad, cond.grad, {k: v.grad for k, v in params.items() if v.grad is not None}
