#!/usr/bin/env python3
"""Mint known-answer vectors for the DDIM hot path from the REFERENCE's own classes.

Runs only in the build container (needs /root/reference; see ref_import.py).  The reference ships
no tests or golden vectors for this path, so these fixtures ARE the pin: each
``*.npz`` holds the OUTPUTS the reference's unmodified Python classes produced on CPU fp32 for
inputs / weights that ``diffusiondepth_amd.synth`` re-creates bit-for-bit from a seed (so inputs
are not stored).  Re-run:   python tests/golden/make_golden.py

Cases (all eval mode, fp32, torch CPU):
  sched.npz      DDIMScheduler tables, timesteps, step() and add_noise() outputs
  denoise_res    ScheduledCNNRefine.forward, scalar t and per-sample t          (Res variant)
  denoise_swin   ScheduledCNNRefine.forward of the Swin variant (UpSample_add fuse)
  loop_res       CNNDDIMPipiline.__call__ T=5 and T=20 (x_T injected by patching torch.randn)
  loop_swin      same for the Swin variant, T=20
  codec          DeepDepthTransformWithUpsampling.t / inv_t (even and odd sizes)
  head_res       full DDIMDepthEstimate_Res.forward (FPN + loop + decoder + ddim_loss), RNG injected
  denoise_bwd_res  autograd of ScheduledCNNRefine.forward (g_x, g_cond, all parameter gradients) for a seeded upstream gradient
  loop_bwd_res   autograd through the whole T-step CNNDDIMPipiline for a seeded dLoss/dx_0
  head_train_res one training-mode forward + backward of the whole reference head (loss, sampled gradients)
  fpn_odd        the head's condition FPN (conv_lateral / conv_up / adaptive_avg_pool2d) on an odd-sized pyramid
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from ref_import import load_reference, load_weights  # noqa: E402
from diffusiondepth_amd import synth  # noqa: E402

CASES = json.load(open(os.path.join(HERE, "cases.json")))


class inject_rng:
    """Make torch.randn / torch.randint return prepared tensors in call order (the reference draws
    x_T with torch.randn (…res.py:277), the loss noise with torch.randn (:203) and the timesteps with
    torch.randint (:207))."""

    def __init__(self, randn_list=(), randint_list=()):
        self.randn_list = [torch.from_numpy(np.asarray(a)) for a in randn_list]
        self.randint_list = [torch.from_numpy(np.asarray(a)) for a in randint_list]

    def __enter__(self):
        self._randn, self._randint = torch.randn, torch.randint

        def randn(*a, **k):
            return self.randn_list.pop(0).clone()

        def randint(*a, **k):
            return self.randint_list.pop(0).clone()

        torch.randn, torch.randint = randn, randint
        return self

    def __exit__(self, *exc):
        torch.randn, torch.randint = self._randn, self._randint
        assert not self.randn_list and not self.randint_list, "unused injected RNG draws"


def t2n(t):
    return t.detach().cpu().numpy().astype(np.float32)


def build(ref, case):
    variant = case.get("variant", "res")
    sd = synth.make_state_dict(case["wseed"], variant, case.get("decoder_gain", 0.05), case.get("decoder_log_scale", 0.0))
    Model = ref.ScheduledCNNRefine if variant == "res" else ref.ScheduledCNNRefineSwin
    model = load_weights(Model(channels_in=256, channels_noise=16).eval(), sd, "model.")
    codec = load_weights(ref.DeepDepthTransformWithUpsampling(hidden=16).eval(), sd, "depth_transform.")
    sched = ref.DDIMScheduler(num_train_timesteps=1000, clip_sample=False)
    return sd, model, codec, sched


def gen_sched(ref):
    c = CASES["sched"]
    s = ref.DDIMScheduler(num_train_timesteps=1000, clip_sample=False)
    out = {"alphas_cumprod": s.alphas_cumprod.numpy().copy(), "betas": s.betas.numpy().copy()}
    rs = np.random.RandomState(c["seed"])
    x = rs.standard_normal(c["shape"]).astype(np.float32)
    eps = np.abs(rs.standard_normal(c["shape"])).astype(np.float32)
    for T in c["T"]:
        s.set_timesteps(T)
        out[f"timesteps_T{T}"] = s.timesteps.numpy().copy()
        prevs = []
        for t in s.timesteps:
            prevs.append(t2n(s.step(torch.from_numpy(eps), t, torch.from_numpy(x), eta=0.0,
                                    use_clipped_model_output=True)["prev_sample"]))
        out[f"step_T{T}"] = np.stack(prevs)
    B = len(c["add_noise_t"])
    x0 = rs.standard_normal((B,) + tuple(c["shape"][1:])).astype(np.float32)
    nz = rs.standard_normal((B,) + tuple(c["shape"][1:])).astype(np.float32)
    s2 = ref.DDIMScheduler(num_train_timesteps=1000, clip_sample=False)
    out["add_noise"] = t2n(s2.add_noise(torch.from_numpy(x0), torch.from_numpy(nz), torch.tensor(c["add_noise_t"])))
    return out


def gen_denoise(ref, name):
    c = CASES[name]
    sd, model, _, _ = build(ref, c)
    inp = synth.make_inputs(c["iseed"], c["B"], c["h"], c["w"], c.get("cond_hw"))
    x, cond = torch.from_numpy(inp["x_T"]), torch.from_numpy(inp["cond"])
    out = {}
    with torch.no_grad():
        out["eps_scalar_t"] = t2n(model(x, torch.tensor(c["t"]), cond, None, None, None))
        out["eps_batch_t"] = t2n(model(x, torch.from_numpy(inp["timesteps"]), cond, None, None, None))
        out["ne_sample0_ch0_8"] = t2n(model.noise_embedding(x))[:1, :8]
    return out


def gen_denoise_bwd(ref, name):
    """Autograd of the reference's ScheduledCNNRefine (the gradients ddim_loss / the depth loss send through one call,
    ...res.py:211 + loss.backward()): g_x, g_cond and every parameter gradient for a seeded upstream gradient."""
    c = CASES[name]
    sd, model, _, _ = build(ref, c)
    inp = synth.make_inputs(c["iseed"], c["B"], c["h"], c["w"], c.get("cond_hw"))
    x = torch.from_numpy(inp["x_T"]).requires_grad_(True)
    cond = torch.from_numpy(inp["cond"]).requires_grad_(True)
    g = torch.from_numpy(np.random.RandomState(c["gseed"]).standard_normal(inp["x_T"].shape).astype(np.float32))
    model.zero_grad()
    eps = model(x, torch.from_numpy(inp["timesteps"]), cond, None, None, None)
    eps.backward(g)
    out = {"eps": t2n(eps), "grad_x": t2n(x.grad), "grad_cond_ch0_8": t2n(cond.grad)[:, :8],
           "grad_cond_chan_sum": cond.grad.double().sum(dim=(0, 2, 3)).numpy()}
    for k, v in model.state_dict(keep_vars=True).items():
        if v.grad is None:
            continue
        gnp = t2n(v.grad)
        if k == "time_embedding.weight":
            rows = sorted(set(int(t) for t in inp["timesteps"]))
            out["grad.model." + k + ".rows"] = np.array(rows, dtype=np.int64)
            out["grad.model." + k] = gnp[rows]
            assert float(np.abs(gnp).sum()) == float(np.abs(gnp[rows]).sum())
        elif gnp.size > 20000:
            # the two 147k-element conv gradients: every 7th element plus two checksums keep the fixture small
            st = 7 if gnp.size <= 200000 else 31
            out["grad.model." + k + f".stride{st}"] = gnp.reshape(-1)[::st].copy()
            out["grad.model." + k + ".sums"] = np.array([gnp.astype(np.float64).sum(), np.abs(gnp.astype(np.float64)).sum()])
        else:
            out["grad.model." + k] = gnp
    return out


def gen_loop_bwd(ref, name):
    """Autograd through the reference's CNNDDIMPipiline (the loop output is not detached in training, ...res.py:124-169):
    dLoss/dcond, dLoss/dx_T and parameter gradients for a seeded dLoss/dx_0."""
    c = CASES[name]
    sd, model, codec, sched = build(ref, c)
    pipe = ref.CNNDDIMPipiline(model, sched)
    inp = synth.make_inputs(c["iseed"], c["B"], c["h"], c["w"])
    cond = torch.from_numpy(inp["cond"]).requires_grad_(True)
    xT = torch.from_numpy(inp["x_T"]).requires_grad_(True)
    g = torch.from_numpy(np.random.RandomState(c["gseed"]).standard_normal(inp["x_T"].shape).astype(np.float32))
    model.zero_grad()
    real_randn = torch.randn
    torch.randn = lambda *a, **k: xT                     # the pipeline's own draw (...res.py:277) returns OUR leaf tensor
    try:
        x0, = pipe(batch_size=c["B"], device=torch.device("cpu"), dtype=torch.float32, shape=(16, c["h"], c["w"]),
                   input_args=(cond, None, None, None), num_inference_steps=c["T"], return_dict=False)
    finally:
        torch.randn = real_randn
    x0.backward(g)
    out = {"x0": t2n(x0), "grad_xT": t2n(xT.grad), "grad_cond_ch0_8": t2n(cond.grad)[:, :8],
           "grad_cond_chan_sum": cond.grad.double().sum(dim=(0, 2, 3)).numpy()}
    for k, v in model.state_dict(keep_vars=True).items():
        if v.grad is None:
            continue
        gnp = t2n(v.grad)
        if k == "time_embedding.weight":
            rows = sorted(set(int(t) for t in sched.timesteps))
            out["grad.model." + k + ".rows"] = np.array(rows, dtype=np.int64)
            out["grad.model." + k] = gnp[rows]
        elif gnp.size > 20000:
            out["grad.model." + k + ".stride7"] = gnp.reshape(-1)[::7].copy()
        else:
            out["grad.model." + k] = gnp
    return out


def gen_loop(ref, name):
    c = CASES[name]
    sd, model, codec, sched = build(ref, c)
    Pipe = ref.CNNDDIMPipiline if c.get("variant", "res") == "res" else ref.CNNDDIMPipilineSwin
    pipe = Pipe(model, sched)
    inp = synth.make_inputs(c["iseed"], c["B"], c["h"], c["w"], c.get("cond_hw"))
    cond = torch.from_numpy(inp["cond"])
    out = {}
    for T in c["T"]:
        with torch.no_grad(), inject_rng([inp["x_T"]]):
            x0, = pipe(batch_size=c["B"], device=torch.device("cpu"), dtype=torch.float32,
                       shape=(16, c["h"], c["w"]), input_args=(cond, None, None, None),
                       num_inference_steps=T, return_dict=False)
            out[f"x0_T{T}"] = t2n(x0)
            out[f"depth_T{T}"] = t2n(codec.inv_t(x0))
    return out


def gen_codec(ref):
    c = CASES["codec"]
    sd, _, codec, _ = build(ref, c)
    out = {}
    for i, (B, H, W) in enumerate(c["sizes"]):
        gt = synth.make_gt_depth(c["iseed"] + i, B, H, W)
        with torch.no_grad():
            lat = codec.t(torch.from_numpy(gt))
            out[f"latent_{i}"] = t2n(lat)
            h, w = synth.latent_hw(H, W)
            z = np.random.RandomState(c["iseed"] + 100 + i).standard_normal((B, 16, h, w)).astype(np.float32) * c["latent_scale"]
            out[f"depth_{i}"] = t2n(codec.inv_t(torch.from_numpy(z)))
    return out


def gen_head(ref):
    c = CASES["head_res"]
    sd = synth.make_state_dict(c["wseed"], "res", c["decoder_gain"], c["decoder_log_scale"])
    fsd = synth.make_fpn_state_dict(c["fseed"])
    head = ref.DDIMDepthEstimate_Res(in_channels=[64, 128, 256, 512], inference_steps=c["T"],
                                     num_train_timesteps=1000, depth_feature_dim=16, loss_cfgs=[]).eval()
    own = head.state_dict()
    full = {}
    full.update({k: torch.from_numpy(v) for k, v in sd.items()})
    full.update({k: torch.from_numpy(v) for k, v in fsd.items()})
    for k in own:
        if k.endswith("num_batches_tracked"):
            full[k] = own[k]
    assert set(full) == set(own), (set(own) ^ set(full))
    head.load_state_dict(full, strict=True)
    B, H, W = c["B"], c["H"], c["W"]
    fp = [torch.from_numpy(f) for f in synth.make_backbone_features(c["iseed"], B, H, W)]
    gt = torch.from_numpy(synth.make_gt_depth(c["iseed"] + 1, B, H, W))
    h, w = synth.latent_hw(H, W)
    inp = synth.make_inputs(c["iseed"] + 2, B, h, w)
    with torch.no_grad(), inject_rng([inp["x_T"], inp["noise"]], [inp["timesteps"]]):
        o = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=False)
    assert set(o) == set(c["output_keys"]), set(o)
    # condition map for the hot-path-only tests
    with torch.no_grad():
        x = None
        for i in range(4):
            f = fp[3 - i]
            cur = head.conv_lateral[3 - i](f)
            if i > 0:
                cur = cur + torch.nn.functional.adaptive_avg_pool2d(head.conv_up[3 - i](x), output_size=cur.shape[-2:])
            x = cur
    return {"pred": t2n(o["pred"]), "pred_init": t2n(o["pred_init"]), "ddim_loss": t2n(o["ddim_loss"]).reshape(1),
            "cond_ch0_4": t2n(x)[:, :4], "cond_sum": np.array([float(x.double().sum())])}


def gen_fpn(ref):
    """Condition FPN of the reference head class on an odd-sized pyramid (both adaptive_avg_pool2d size fixes are active)."""
    c = CASES["fpn_odd"]
    fsd = synth.make_fpn_state_dict(c["fseed"])
    head = ref.DDIMDepthEstimate_Res(in_channels=[64, 128, 256, 512], inference_steps=2, num_train_timesteps=1000,
                                     depth_feature_dim=16, loss_cfgs=[]).eval()
    own = head.state_dict()
    head.load_state_dict({k: (torch.from_numpy(fsd[k]) if k in fsd else own[k]) for k in own}, strict=True)
    fp = [torch.from_numpy(f) for f in synth.make_backbone_features(c["iseed"], c["B"], c["H"], c["W"])]
    with torch.no_grad():
        x = None
        for i in range(4):                      # the loop of ...res.py:108-118, module calls are the reference's own
            cur = head.conv_lateral[3 - i](fp[3 - i])
            if i > 0:
                cur = cur + torch.nn.functional.adaptive_avg_pool2d(head.conv_up[3 - i](x), output_size=cur.shape[-2:])
            x = cur
    return {"cond_ch0_8": t2n(x)[:, :8], "cond_chan_sum": x.double().sum(dim=(0, 2, 3)).numpy(),
            "shape": np.array(x.shape, dtype=np.int64)}


def gen_head_train(ref):
    """One training-mode forward + backward of the reference head (BatchNorm on batch statistics in the FPN and the codec,
    autograd through the T-step loop and ddim_loss; RNG draws injected): loss value and a sample of parameter gradients."""
    c = CASES["head_train_res"]
    sd = synth.make_state_dict(c["wseed"], "res", c["decoder_gain"], c["decoder_log_scale"])
    fsd = synth.make_fpn_state_dict(c["fseed"])
    head = ref.DDIMDepthEstimate_Res(in_channels=[64, 128, 256, 512], inference_steps=c["T"], num_train_timesteps=1000,
                                     depth_feature_dim=16, loss_cfgs=[]).train()
    own = head.state_dict()
    full = {k: torch.from_numpy(v) for k, v in {**sd, **fsd}.items()}
    for k in own:
        if k.endswith("num_batches_tracked"):
            full[k] = own[k]
    head.load_state_dict(full, strict=True)
    B, H, W = c["B"], c["H"], c["W"]
    fp = [torch.from_numpy(f).requires_grad_(True) for f in synth.make_backbone_features(c["iseed"], B, H, W)]
    gt = torch.from_numpy(synth.make_gt_depth(c["iseed"] + 1, B, H, W))
    h, w = synth.latent_hw(H, W)
    inp = synth.make_inputs(c["iseed"] + 2, B, h, w)
    with inject_rng([inp["x_T"], inp["noise"]], [inp["timesteps"]]):
        o = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=False)
        loss = (o["pred"] - gt).abs().mean() + o["ddim_loss"]
    loss.backward()
    out = {"loss": np.array([float(loss)]), "ddim_loss": np.array([float(o["ddim_loss"])]), "pred": t2n(o["pred"]),
           "grad_fp3": t2n(fp[3].grad), "grad_fp0_ch0_4": t2n(fp[0].grad)[:, :4]}
    for k in c["grad_keys"]:
        p = dict(head.named_parameters())[k]
        gnp = t2n(p.grad)
        out["grad." + k] = gnp.reshape(-1)[::c["grad_stride"]].copy() if gnp.size > 5000 else gnp
    return out


def main():
    ref = load_reference()
    torch.manual_seed(0)
    torch.set_num_threads(8)
    gens = {
        "sched": lambda: gen_sched(ref),
        "denoise_res": lambda: gen_denoise(ref, "denoise_res"),
        "denoise_swin": lambda: gen_denoise(ref, "denoise_swin"),
        "denoise_bwd_res": lambda: gen_denoise_bwd(ref, "denoise_bwd_res"),
        "loop_bwd_res": lambda: gen_loop_bwd(ref, "loop_bwd_res"),
        "denoise_bwd_swin": lambda: gen_denoise_bwd(ref, "denoise_bwd_swin"),
        "loop_res": lambda: gen_loop(ref, "loop_res"),
        "loop_res_far": lambda: gen_loop(ref, "loop_res_far"),
        "loop_swin": lambda: gen_loop(ref, "loop_swin"),
        "codec": lambda: gen_codec(ref),
        "head_res": lambda: gen_head(ref),
        "fpn_odd": lambda: gen_fpn(ref),
        "head_train_res": lambda: gen_head_train(ref),
    }
    only = sys.argv[1:]
    for name, fn in gens.items():
        if only and name not in only:
            continue
        out = fn()
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB  keys={sorted(out)}")


if __name__ == "__main__":
    main()
