"""CPU oracle for the DDIM denoise hot path  --  TEST INFRASTRUCTURE ONLY.

A dependency-free NumPy restatement (default float64) of the reference algorithm on the hot
path of duanyiqun/DiffusionDepth.  It is the checker the HIP kernels are compared against.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; the product package ``diffusiondepth_amd`` never does.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function here against
known-answer vectors minted from the reference's own classes run on CPU
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).  The reference itself ships no
tests or golden vectors for this path.

Layout convention: NCHW float arrays, exactly like the reference tensors.
Every function cites the reference lines it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import numpy as np

GN_EPS = 1e-5   # torch.nn.GroupNorm default eps
BN_EPS = 1e-5   # torch.nn.BatchNorm2d default eps


# --------------------------------------------------------------------------------------------
# torch.nn primitives restated (semantics of the stock torch ops the reference calls)
# --------------------------------------------------------------------------------------------
def conv2d(x, w, b=None, stride=1, pad=1):
    """nn.Conv2d (cross-correlation, zero padding).  x (B,Ci,H,W), w (Co,Ci,kh,kw)."""
    B, Ci, H, W = x.shape
    Co, Ci2, kh, kw = w.shape
    assert Ci == Ci2
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    xp = np.zeros((B, Ci, H + 2 * pad, W + 2 * pad), dtype=x.dtype)
    xp[:, :, pad:pad + H, pad:pad + W] = x
    out = np.zeros((B, Co, Ho, Wo), dtype=np.result_type(x.dtype, w.dtype))
    for dy in range(kh):
        for dx in range(kw):
            patch = xp[:, :, dy:dy + (Ho - 1) * stride + 1:stride, dx:dx + (Wo - 1) * stride + 1:stride]
            # (Co,Ci) x (B,Ci,Ho*Wo) -> (B,Co,Ho*Wo)
            out += np.matmul(w[:, :, dy, dx], patch.reshape(B, Ci, Ho * Wo)).reshape(B, Co, Ho, Wo)
    if b is not None:
        out += b.reshape(1, Co, 1, 1)
    return out


def conv_transpose2d(x, w, b=None, stride=2, pad=1):
    """nn.ConvTranspose2d.  x (B,Ci,H,W), w (Ci,Co,kh,kw); out size (H-1)*s - 2p + k."""
    B, Ci, H, W = x.shape
    Ci2, Co, kh, kw = w.shape
    assert Ci == Ci2
    Ho = (H - 1) * stride - 2 * pad + kh
    Wo = (W - 1) * stride - 2 * pad + kw
    full = np.zeros((B, Co, (H - 1) * stride + kh, (W - 1) * stride + kw), dtype=np.result_type(x.dtype, w.dtype))
    xf = x.reshape(B, Ci, H * W)
    for ky in range(kh):
        for kx in range(kw):
            contrib = np.matmul(w[:, :, ky, kx].T, xf).reshape(B, Co, H, W)
            full[:, :, ky:ky + (H - 1) * stride + 1:stride, kx:kx + (W - 1) * stride + 1:stride] += contrib
    out = full[:, :, pad:pad + Ho, pad:pad + Wo]
    if b is not None:
        out = out + b.reshape(1, Co, 1, 1)
    return out


def group_norm(x, groups, gamma, beta, eps=GN_EPS):
    """nn.GroupNorm: per-sample, per-group mean / biased variance over (C/G, H, W)."""
    B, C, H, W = x.shape
    xg = x.reshape(B, groups, -1)
    mean = xg.mean(axis=2, keepdims=True)
    var = xg.var(axis=2, keepdims=True)          # biased (ddof=0), as torch
    y = ((xg - mean) / np.sqrt(var + eps)).reshape(B, C, H, W)
    return y * gamma.reshape(1, C, 1, 1) + beta.reshape(1, C, 1, 1)


def batch_norm_eval(x, gamma, beta, mean, var, eps=BN_EPS):
    """nn.BatchNorm2d in eval mode (running statistics)."""
    s = gamma / np.sqrt(var + eps)
    return x * s.reshape(1, -1, 1, 1) + (beta - mean * s).reshape(1, -1, 1, 1)


def relu(x):
    return np.maximum(x, 0)


def leaky_relu(x, slope=0.2):
    return np.where(x >= 0, x, x * slope)


def bilinear_align_corners(x, out_h, out_w):
    """F.interpolate(mode='bilinear', align_corners=True)."""
    B, C, H, W = x.shape
    if (H, W) == (out_h, out_w):
        return x.copy()
    ys = np.arange(out_h) * ((H - 1) / (out_h - 1) if out_h > 1 else 0.0)
    xs = np.arange(out_w) * ((W - 1) / (out_w - 1) if out_w > 1 else 0.0)
    y0 = np.clip(np.floor(ys).astype(np.int64), 0, H - 1)
    x0 = np.clip(np.floor(xs).astype(np.int64), 0, W - 1)
    y1 = np.minimum(y0 + 1, H - 1)
    x1 = np.minimum(x0 + 1, W - 1)
    wy = (ys - y0).reshape(1, 1, -1, 1)
    wx = (xs - x0).reshape(1, 1, 1, -1)
    top = x[:, :, y0][:, :, :, x0] * (1 - wx) + x[:, :, y0][:, :, :, x1] * wx
    bot = x[:, :, y1][:, :, :, x0] * (1 - wx) + x[:, :, y1][:, :, :, x1] * wx
    return top * (1 - wy) + bot * wy


# --------------------------------------------------------------------------------------------
# DDIM scheduler  (reference src/model/diffusers/schedulers/scheduling_ddim.py)
# --------------------------------------------------------------------------------------------
def _torch_linspace_f32(start, end, steps):
    """Bit-faithful restatement of torch.linspace(dtype=float32) on CPU (checked against torch in
    tests/test_oracle_golden.py): step is computed in fp32; the first half is start + step*i, the
    second half end - step*(steps-1-i), each evaluated as ONE fused multiply-add (single rounding)."""
    start = np.float32(start)
    end = np.float32(end)
    step = np.float64(np.float32((end - start) / np.float32(steps - 1)))
    i = np.arange(steps)
    half = steps // 2
    lo = (np.float64(start) + step * i).astype(np.float32)              # exact in f64 -> one rounding
    hi = (np.float64(end) - step * (steps - 1 - i)).astype(np.float32)
    return np.where(i < half, lo, hi).astype(np.float32)


class DDIMScheduleOracle:
    """linear-beta, epsilon-prediction, eta = 0 DDIM (scheduling_ddim.py:107-157, 215-229, 231-353)."""

    def __init__(self, num_train_timesteps=1000, beta_start=1e-4, beta_end=0.02):
        self.num_train_timesteps = num_train_timesteps
        # scheduling_ddim.py:131  torch.linspace(..., dtype=float32)
        self.betas = _torch_linspace_f32(beta_start, beta_end, num_train_timesteps)
        # scheduling_ddim.py:143-144  alphas = 1 - betas (fp32); torch.cumprod on CPU accumulates a
        # float tensor in double and rounds each prefix product to fp32
        alphas = (np.float32(1.0) - self.betas).astype(np.float32)
        self.alphas_cumprod = np.cumprod(alphas.astype(np.float64)).astype(np.float32)
        self.final_alpha_cumprod = np.float32(1.0)      # scheduling_ddim.py:150 (set_alpha_to_one=True)
        self.timesteps = None
        self.num_inference_steps = None

    def set_timesteps(self, T):
        # scheduling_ddim.py:215-229
        self.num_inference_steps = T
        ratio = self.num_train_timesteps // T
        self.timesteps = (np.arange(0, T) * ratio).round()[::-1].copy().astype(np.int64)
        return self.timesteps

    def alpha_pair(self, t):
        # scheduling_ddim.py:285-289
        prev = int(t) - self.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[int(t)]
        a_prev = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        return a_t, a_prev

    def step(self, eps, t, x, dtype=np.float64):
        """scheduling_ddim.py:285-326 with eta=0, clip_sample=False, use_clipped_model_output=True,
        written out literally (not the closed form) in ``dtype`` arithmetic."""
        a_t, a_prev = self.alpha_pair(t)
        a_t = dtype(a_t)
        a_prev = dtype(a_prev)
        beta_t = 1 - a_t
        x0 = (x - beta_t ** 0.5 * eps) / a_t ** 0.5                     # :296
        eps2 = (x - a_t ** 0.5 * x0) / beta_t ** 0.5                    # :318-320
        direction = (1 - a_prev - 0.0) ** 0.5 * eps2                    # :323  (std_dev_t = 0)
        return a_prev ** 0.5 * x0 + direction                           # :326

    def coeffs(self, t):
        """Closed form of step(): x_prev = c1*x + c2*eps, float64."""
        a_t, a_prev = (np.float64(v) for v in self.alpha_pair(t))
        c1 = np.sqrt(a_prev / a_t)
        c2 = np.sqrt(1 - a_prev) - np.sqrt(a_prev * (1 - a_t) / a_t)
        return c1, c2

    def add_noise(self, x0, noise, timesteps, dtype=np.float64):
        """q_sample, scheduling_ddim.py:355-376 (table is fp32, broadcast per sample)."""
        a = self.alphas_cumprod[np.asarray(timesteps, dtype=np.int64)].astype(dtype)
        sa = (a ** 0.5).reshape(-1, 1, 1, 1)
        sb = ((1 - a) ** 0.5).reshape(-1, 1, 1, 1)
        return sa * x0 + sb * noise


# --------------------------------------------------------------------------------------------
# epsilon network  (reference src/model/head/ddim_depth_estimate_res.py:300-344 and
#                   src/model/head/ddim_depth_estimate_res_swin_addHAHI.py:321-333,336-382)
# --------------------------------------------------------------------------------------------
def _cast_sd(sd, dtype):
    return {k: np.asarray(v).astype(dtype) for k, v in sd.items()}


def noise_embedding(sd, x):
    """…res.py:303-311: Conv3x3(16->64)+GN(4)+ReLU + Conv3x3(64->256)+GN(4)+ReLU."""
    y = conv2d(x, sd["model.noise_embedding.0.weight"], sd["model.noise_embedding.0.bias"])
    y = relu(group_norm(y, 4, sd["model.noise_embedding.1.weight"], sd["model.noise_embedding.1.bias"]))
    y = conv2d(y, sd["model.noise_embedding.3.weight"], sd["model.noise_embedding.3.bias"])
    y = relu(group_norm(y, 4, sd["model.noise_embedding.4.weight"], sd["model.noise_embedding.4.bias"]))
    return y


def pred_head(sd, f):
    """…res.py:315-322: Conv3x3(256->64)+GN(4)+ReLU + Conv3x3(64->16)+GN(4)+ReLU  (eps >= 0)."""
    y = conv2d(f, sd["model.pred.0.weight"], sd["model.pred.0.bias"])
    y = relu(group_norm(y, 4, sd["model.pred.1.weight"], sd["model.pred.1.bias"]))
    y = conv2d(y, sd["model.pred.3.weight"], sd["model.pred.3.bias"])
    y = relu(group_norm(y, 4, sd["model.pred.4.weight"], sd["model.pred.4.bias"]))
    return y


def denoiser_forward(sd, x_t, t, cond, variant="res", dtype=np.float64, return_intermediates=False):
    """ScheduledCNNRefine.forward.  t: python int / 0-d (loop) or (B,) int array (ddim_loss).

    res  (…res.py:324-344):            f = cond + E[t] + NE(x_t);  eps = PRED(f)
    swin (…swin_addHAHI.py:364-382):   f = convB(convA(bilinear_up(cond + E[t]) + NE(x_t)));  eps = PRED(f)
    """
    sd = _cast_sd(sd, dtype)
    x_t = np.asarray(x_t, dtype=dtype)
    cond = np.asarray(cond, dtype=dtype)
    B = x_t.shape[0]
    t = np.asarray(t, dtype=np.int64)
    emb = sd["model.time_embedding.weight"][t]          # (256,) or (B,256)
    emb = emb.reshape((1 if emb.ndim == 1 else B), -1, 1, 1)
    feat = cond + emb                                    # …res.py:330/335
    ne = noise_embedding(sd, x_t)
    if variant == "res":
        f = feat + ne                                    # …res.py:340
    elif variant == "swin":
        up = bilinear_align_corners(feat, ne.shape[2], ne.shape[3])       # …swin_addHAHI.py:332
        f = conv2d(up + ne, sd["model.upsample_fuse.convA.conv.weight"], sd["model.upsample_fuse.convA.conv.bias"])
        f = conv2d(f, sd["model.upsample_fuse.convB.conv.weight"], sd["model.upsample_fuse.convB.conv.bias"])
    else:
        raise ValueError(variant)
    eps = pred_head(sd, f)                               # …res.py:342
    if return_intermediates:
        return eps, {"ne": ne, "f": f}
    return eps


def ddim_loop(sd, x_T, cond, T=20, variant="res", dtype=np.float64, literal_step=True, return_traj=False,
              num_train_timesteps=1000):
    """CNNDDIMPipiline.__call__ (…res.py:248-297) with x_T injected instead of torch.randn (:277)."""
    sch = DDIMScheduleOracle(num_train_timesteps)
    x = np.asarray(x_T, dtype=dtype)
    traj = []
    for t in sch.set_timesteps(T):                                       # :280-282
        eps = denoiser_forward(sd, x, int(t), cond, variant, dtype)      # :285
        if literal_step:
            x = sch.step(eps, int(t), x, dtype)                          # :290-292
        else:
            c1, c2 = sch.coeffs(int(t))
            x = dtype(c1) * x + dtype(c2) * eps
        if return_traj:
            traj.append(x.copy())
    return (x, traj) if return_traj else x


def ddim_loss(sd, x0_latent, cond, noise, timesteps, variant="res", dtype=np.float64):
    """DDIMDepthEstimate_*.ddim_loss (…res.py:201-217) with noise / timesteps injected."""
    sch = DDIMScheduleOracle()
    noisy = sch.add_noise(np.asarray(x0_latent, dtype=dtype), np.asarray(noise, dtype=dtype), timesteps, dtype)
    pred = denoiser_forward(sd, noisy, np.asarray(timesteps), cond, variant, dtype)
    return np.mean((pred - np.asarray(noise, dtype=dtype)) ** 2), pred, noisy


# --------------------------------------------------------------------------------------------
# latent encoder / decoder  (reference src/model/ops/depth_transform.py:10-35, src/model/common.py:45-60)
# --------------------------------------------------------------------------------------------
def _bn_args(sd, prefix):
    return (sd[prefix + ".weight"], sd[prefix + ".bias"], sd[prefix + ".running_mean"], sd[prefix + ".running_var"])


def encode(sd, depth, dtype=np.float64):
    """DeepDepthTransformWithUpsampling.t (eval-mode BN): depth_transform.py:15-19,29-31."""
    sd = _cast_sd(sd, dtype)
    p = "depth_transform.conv_transform."
    y = conv2d(np.asarray(depth, dtype=dtype), sd[p + "0.0.weight"], None, stride=2, pad=1)
    y = leaky_relu(batch_norm_eval(y, *_bn_args(sd, p + "0.1")), 0.2)     # common.py:53-56
    y = conv2d(y, sd[p + "1.0.weight"], None, stride=1, pad=1)
    y = batch_norm_eval(y, *_bn_args(sd, p + "1.1"))
    return np.tanh(y)


def decode_logit(sd, latent, dtype=np.float64):
    """conv_inv_transform up to (excluding) the Sigmoid: depth_transform.py:20-25."""
    sd = _cast_sd(sd, dtype)
    p = "depth_transform.conv_inv_transform."
    y = conv_transpose2d(np.asarray(latent, dtype=dtype), sd[p + "0.weight"], sd[p + "0.bias"], stride=2, pad=1)
    y = relu(batch_norm_eval(y, *_bn_args(sd, p + "1")))
    return conv2d(y, sd[p + "3.0.weight"], sd[p + "3.0.bias"], stride=1, pad=1)


def decode(sd, latent, eps=1e-6, dtype=np.float64):
    """DeepDepthTransformWithUpsampling.inv_t: 1/clamp(sigmoid(z), eps) - 1  (depth_transform.py:33-35)."""
    z = decode_logit(sd, latent, dtype)
    s = 1.0 / (1.0 + np.exp(-z))
    return 1.0 / np.maximum(s, eps) - 1.0


def adaptive_avg_pool2d(x, out_h, out_w):
    """F.adaptive_avg_pool2d: output (i, j) averages input rows floor(i*H/oh) .. ceil((i+1)*H/oh)-1 and the same for
    columns (ATen AdaptiveAveragePooling.cpp start_index / end_index)."""
    B, C, H, W = x.shape
    if (H, W) == (out_h, out_w):
        return x
    out = np.empty((B, C, out_h, out_w), dtype=x.dtype)
    for i in range(out_h):
        ys, ye = (i * H) // out_h, -((-(i + 1) * H) // out_h)
        for j in range(out_w):
            xs, xe = (j * W) // out_w, -((-(j + 1) * W) // out_w)
            out[:, :, i, j] = x[:, :, ys:ye, xs:xe].mean(axis=(2, 3))
    return out


def fpn_aggregate(fsd, fp, dtype=np.float64):
    """Condition aggregation of DDIMDepthEstimate_Res.forward (reference src/model/head/ddim_depth_estimate_res.py:108-118):
    top-down over the pyramid, x = conv_lateral[i](f_i) [+ adaptive_avg_pool2d(conv_up[i](x_prev), size of x)], with
    conv_lateral = Conv3x3(bias=False)+BN(eval)+ReLU (...res.py:60-69) and conv_up = ConvTranspose2d(k2, s2, bias=False)
    +BN(eval)+ReLU (...res.py:71-84).  fsd: state-dict entries "conv_lateral.*" / "conv_up.*"; fp: list of 4 NCHW maps."""
    sd = _cast_sd(fsd, dtype)
    n = len(fp)
    x = None
    for i in range(n):
        lvl = n - i - 1
        f = np.asarray(fp[lvl], dtype=dtype)
        cur = relu(batch_norm_eval(conv2d(f, sd[f"conv_lateral.{lvl}.0.weight"]), *_bn_args(sd, f"conv_lateral.{lvl}.1")))
        if i > 0:
            up = conv_transpose2d(x, sd[f"conv_up.{lvl}.0.weight"], None, stride=2, pad=0)
            up = relu(batch_norm_eval(up, *_bn_args(sd, f"conv_up.{lvl}.1")))
            cur = cur + adaptive_avg_pool2d(up, cur.shape[2], cur.shape[3])
        x = cur
    return x


def head_hot_path(sd, gt_depth, cond, x_T, noise, timesteps, T=20, variant="res", dtype=np.float64):
    """The hot-path part of DDIMDepthEstimate_Res.forward (…res.py:102,124-176) given the condition
    map (the FPN stays outside the hot path): encoder -> T-step loop -> decoder -> ddim_loss."""
    gt_map_t = encode(sd, gt_depth, dtype)                                   # :102
    assert gt_map_t.shape[-3:] == np.asarray(x_T).shape[-3:]
    refined_t = ddim_loop(sd, x_T, cond, T, variant, dtype)                  # :124-138
    pred = decode(sd, refined_t, dtype=dtype)                                # :140
    loss, _, _ = ddim_loss(sd, refined_t, cond, noise, timesteps, variant, dtype)   # :159-169
    return {"pred": pred, "pred_init": gt_map_t, "blur_depth_t": gt_map_t, "gt_map_t": gt_map_t,
            "ddim_loss": loss, "refined_depth_t": refined_t}
