#!/usr/bin/env python3
"""CPU experiment for DESIGN.md section 7 item 0: how much accuracy would a Winograd F(2x2,3x3) conv2 / conv3 cost in the 16-bit
operand modes?  Emulates the Res denoiser loop the way the HIP kernels compute it -- fp32 state / accumulators / GroupNorm, activations
between layers and MFMA operands rounded to bf16 or f16 -- once with direct 3x3 convolutions and once with conv2 / conv3 in Winograd
form (input transform V = B^T d B rounded to 16 bit, pre-transformed weights U = G g G^T rounded to 16 bit, fp32 products, output
transform in fp32), and reports the decoded-depth error of each against the fp32 reference loop (oracle/torch_cpu_port.py).

    python tools/winograd_numerics.py [--h 24 --w 40 --T 20 --seeds 3]

Test infrastructure (imports oracle/); nothing here is part of the product."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffusiondepth_amd import synth  # noqa: E402
from oracle import torch_cpu_port as P  # noqa: E402

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def rnd(x, dt):
    return x if dt is None else x.to(dt).float()


def conv_direct(a, w, b, dt):
    return F.conv2d(rnd(a, dt), rnd(w, dt), b, padding=1)


NATIVE_ADDS = False       # True: the input transform's additions run in the 16-bit type itself (v_pk_add_f16: a rounding after EVERY add)


def _bt_d_b_native(tiles, dt):
    """B^T d B with every addition rounded to dt (what packed 16-bit VALU adds would do); tiles (...,4,4) already in dt."""
    r = lambda x: rnd(x, dt)
    d0, d1, d2, d3 = tiles[..., 0, :], tiles[..., 1, :], tiles[..., 2, :], tiles[..., 3, :]
    t = torch.stack([r(d0 - d2), r(d1 + d2), r(d2 - d1), r(d1 - d3)], dim=-2)
    c0, c1, c2, c3 = t[..., 0], t[..., 1], t[..., 2], t[..., 3]
    return torch.stack([r(c0 - c2), r(c1 + c2), r(c2 - c1), r(c1 - c3)], dim=-1)


def conv_winograd(a, w, b, dt):
    """F(2x2,3x3): a (B,C,H,W) fp32 (already the conv's input values), w (Co,C,3,3).  16-bit rounding is applied to V and U."""
    Bn, C, H, W = a.shape
    Hp, Wp = H + (H % 2), W + (W % 2)
    x = F.pad(a, (1, 1 + Wp - W, 1, 1 + Hp - H))                                  # zero padding of the conv + even size
    tiles = x.unfold(2, 4, 2).unfold(3, 4, 2)                                      # (B,C,Hp/2,Wp/2,4,4)
    if NATIVE_ADDS and dt is not None:
        V = _bt_d_b_native(rnd(tiles, dt), dt)
    else:
        V = rnd(torch.einsum("ij,bchwjk,lk->bchwil", BT, rnd(tiles, dt), BT), dt)  # the MFMA operand is V rounded; d itself is 16-bit too
    U = rnd(torch.einsum("ij,ocjk,lk->ocil", G, w, G), dt)                         # pre-transformed in fp32, then rounded
    M = torch.einsum("ocil,bchwil->bohwil", U, V)                                  # fp32 accumulation over channels
    Y = torch.einsum("ij,bohwjk,lk->bohwil", AT, M, AT)                            # (B,Co,Hp/2,Wp/2,2,2)
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(Bn, w.shape[0], Hp, Wp)[:, :, :H, :W]
    return y + b.view(1, -1, 1, 1)


def denoiser_emul(sd, x, t, cond, dt, wino):
    """The four fused kernels of the Res denoiser: conv outputs are stored in `dt`, GroupNorm statistics come from the fp32 values."""
    conv23 = conv_winograd if wino else conv_direct
    gn = lambda y, p: F.relu(F.group_norm(y, 4, sd[p + ".weight"], sd[p + ".bias"]))
    emb = F.embedding(torch.as_tensor(t, dtype=torch.long), sd["model.time_embedding.weight"])[..., None, None]

    def layer(inp, conv, wk, gk):
        y = conv(inp, sd[wk + ".weight"], sd[wk + ".bias"], dt)                  # fp32 accumulators + bias
        st = F.group_norm(y, 4, sd[gk + ".weight"], sd[gk + ".bias"])             # statistics from the fp32 values ...
        mean = y.view(y.shape[0], 4, -1).mean(-1)
        var = y.view(y.shape[0], 4, -1).var(-1, unbiased=False)
        ys = rnd(y, dt)                                                           # ... the stored tensor is 16-bit
        C = y.shape[1]
        yn = (ys.view(y.shape[0], 4, -1) - mean[..., None]) / torch.sqrt(var[..., None] + 1e-5)
        yn = yn.view_as(y) * sd[gk + ".weight"].view(1, C, 1, 1) + sd[gk + ".bias"].view(1, C, 1, 1)
        del st
        return F.relu(yn)

    a1 = layer(x, conv_direct, "model.noise_embedding.0", "model.noise_embedding.1")      # x is fp32 state rounded inside conv_direct
    a2 = layer(a1, conv23, "model.noise_embedding.3", "model.noise_embedding.4")
    f = a2 + rnd(cond, dt) + emb
    a3 = layer(f, conv23, "model.pred.0", "model.pred.1")
    y4 = conv_direct(a3, sd["model.pred.3.weight"], sd["model.pred.3.bias"], dt)           # conv4 output stays fp32
    return F.relu(F.group_norm(y4, 4, sd["model.pred.4.weight"], sd["model.pred.4.bias"]))


@torch.no_grad()
def loop(sd, x_T, cond, T, dt, wino):
    acp = P.make_alphas_cumprod(1000)
    x = torch.as_tensor(x_T)
    cond = torch.as_tensor(cond)
    for t in P.timesteps(T, 1000):
        eps = denoiser_emul(sd, x, int(t), cond, dt, wino)
        x = P.ddim_step(acp, eps, int(t), x, 1000 // T)
    return x


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--h", type=int, default=24)
    ap.add_argument("--w", type=int, default=40)
    ap.add_argument("--T", type=int, default=20)
    ap.add_argument("--seeds", type=int, default=3)
    a = ap.parse_args()
    torch.set_num_threads(8)
    rows = {}
    for s in range(a.seeds):
        sd = P.to_torch_sd(synth.make_state_dict(7240 + s))
        inp = synth.make_inputs(100 + s, 1, a.h, a.w)
        ref = P.decode(sd, P.ddim_loop(sd, inp["x_T"], inp["cond"], a.T))
        chk = P.decode(sd, loop(sd, inp["x_T"], inp["cond"], a.T, None, True))            # Winograd in fp32: must be ~exact
        rows.setdefault("fp32 winograd (sanity)", []).append(float((chk - ref).pow(2).mean().sqrt()))
        for name, dt in (("bf16", torch.bfloat16), ("f16", torch.float16)):
            for wino in (False, True):
                d = P.decode(sd, loop(sd, inp["x_T"], inp["cond"], a.T, dt, wino))
                rows.setdefault(f"{name} {'winograd conv2/conv3' if wino else 'direct'}", []).append(float((d - ref).pow(2).mean().sqrt()))
        global NATIVE_ADDS
        NATIVE_ADDS = True
        d = P.decode(sd, loop(sd, inp["x_T"], inp["cond"], a.T, torch.float16, True))
        rows.setdefault("f16 winograd, transform adds in f16", []).append(float((d - ref).pow(2).mean().sqrt()))
        NATIVE_ADDS = False
        print(f"seed {s}: depth range {float(ref.min()):.2f}..{float(ref.max()):.2f}", flush=True)
    print(f"\nlatent {a.h}x{a.w}, T={a.T}: depth RMSE vs the fp32 reference loop (mean over {a.seeds} weight / input seeds)")
    for k, v in rows.items():
        print(f"  {k:32s} {np.mean(v):.3e}   (per seed: {', '.join(f'{x:.2e}' for x in v)})")


if __name__ == "__main__":
    main()
