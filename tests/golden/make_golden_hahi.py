#!/usr/bin/env python3
"""Mint known-answer vectors for the head variants that round 1 adds on top of Res / Swin_ADD, from the REFERENCE's own classes:

  head_swin_hahi.npz   DDIMDepthEstimate_Swin_ADDHAHI.forward -- the head of the reference's headline configuration (README.md:215):
                       HAHIHeteroNeck (attention off, src/model/necks/hahi.py) -> condition FPN -> 20-step loop -> decoder -> ddim_loss
  head_mpvit_hahi.npz  DDIMDepthEstimate_MPVIT_ADDHAHI.forward (pyramid widths 128/216/288/288, odd-sized maps: both adaptive_avg_pool2d fixes active)
  head_res_vis.npz     DDIMDepthEstimate_ResVis.forward -- 'pred_inter' = every intermediate sample of the loop, decoded
                       (src/model/head/ddim_depth_estimate_res_vis.py:124,141-143,177)

Build container only.  Beyond tests/golden/ref_import.py's stand-ins this needs, for hahi.py:7-12:
  mmcv.cnn.ConvModule with norm_cfg=BN / act_cfg=ReLU   -> conv (bias-free) + ``bn`` + ``activate``: mmcv's own attribute names
  mmcv.cnn.xavier_init, mmcv.runner.auto_fp16           -> only used by init_weights() / decorators (not called here)
  mmcv.cnn.bricks.transformer.build_positional_encoding -> zeros of (B, 2*num_feats, H, W): the encoding only feeds the attention
                                                           modules and ``reference_points``, whose results are discarded when
                                                           cross_att = self_att = False (hahi.py:211-247)
  mmcv.ops.multi_scale_deform_attn.MultiScaleDeformableAttention -> a parameter container with mmcv's parameter names
  model.ops.resize                                      -> never called (scales == 1)
The stand-ins define containers and parameter NAMES (mmcv's, un-vendored: requirements.txt pins mmcv-full==1.3.13); every arithmetic
operation that reaches the stored outputs is the reference's forward code plus stock torch.nn.  Weights / inputs come from
diffusiondepth_amd.synth seeds (tests/golden/cases.json) and are not stored.   Re-run: python tests/golden/make_golden_hahi.py
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import ref_import  # noqa: E402
from ref_import import _load_as, load_reference  # noqa: E402
from make_golden import inject_rng  # noqa: E402
from diffusiondepth_amd import synth  # noqa: E402

CASES = json.load(open(os.path.join(HERE, "cases.json")))
t2n = lambda t: t.detach().cpu().numpy()


class _ConvModuleNA(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, conv_cfg=None, norm_cfg=None, act_cfg=None, **kw):
        super().__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=norm_cfg is None)
        if norm_cfg is not None:
            assert norm_cfg["type"] == "BN"
            self.bn = nn.BatchNorm2d(out_channels)
        if act_cfg is not None:
            assert act_cfg["type"] == "ReLU"
            self.activate = nn.ReLU(inplace=True)
        self._n, self._a = norm_cfg is not None, act_cfg is not None

    def forward(self, x):
        x = self.conv(x)
        if self._n:
            x = self.bn(x)
        return self.activate(x) if self._a else x


class _MSDA(nn.Module):
    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, **kw):
        super().__init__()
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)

    def forward(self, *a, **k):
        raise RuntimeError("attention is off in every DiffusionDepth head")


class _ZeroPos(nn.Module):
    def __init__(self, num_feats):
        super().__init__()
        self.num_feats = num_feats

    def forward(self, mask):
        B, H, W = mask.shape
        return torch.zeros(B, 2 * self.num_feats, H, W, device=mask.device)


def load_hahi_reference():
    ns = load_reference()
    m = sys.modules
    m["mmcv.cnn"].ConvModule = _ConvModuleNA          # superset of ref_import's plain-conv stand-in
    m["mmcv.cnn"].xavier_init = lambda *a, **k: None
    m["mmcv.runner"].auto_fp16 = lambda *a, **k: (lambda f: f)
    for name in ("mmcv.cnn.bricks", "mmcv.cnn.bricks.transformer", "mmcv.ops", "mmcv.ops.multi_scale_deform_attn"):
        m[name] = types.ModuleType(name)
    m["mmcv.cnn.bricks.transformer"].build_positional_encoding = lambda cfg: _ZeroPos(cfg["num_feats"])
    m["mmcv.ops.multi_scale_deform_attn"].MultiScaleDeformableAttention = _MSDA
    m["model.ops"].resize = None
    necks = types.ModuleType("model.necks")
    necks.__path__ = [os.path.join(ref_import.REF_SRC, "model", "necks")]
    m["model.necks"] = necks
    hahi = _load_as("model.necks.hahi", "model/necks/hahi.py")
    head = _load_as("model.head.ddim_depth_estimate_res_swin_addHAHI", "model/head/ddim_depth_estimate_res_swin_addHAHI.py")
    vis = _load_as("model.head.ddim_depth_estimate_res_vis", "model/head/ddim_depth_estimate_res_vis.py")
    mpvit = _load_as("model.head.ddim_depth_estimate_res_mpvit_HAHI", "model/head/ddim_depth_estimate_res_mpvit_HAHI.py")
    return ns, mpvit.DDIMDepthEstimate_MPVIT_ADDHAHI, head.DDIMDepthEstimate_Swin_ADDHAHI, vis.DDIMDepthEstimate_ResVis


def _load(head, *sds):
    own = head.state_dict()
    full = {}
    for sd in sds:
        full.update({k: torch.from_numpy(v) for k, v in sd.items()})
    for k in own:
        if k.endswith("num_batches_tracked"):
            full[k] = own[k]
    assert set(full) == set(own), sorted(set(own) ^ set(full))[:10]
    head.load_state_dict(full, strict=True)


def gen_head_swin_hahi(HeadCls, case="head_swin_hahi", chans=(192, 384, 768, 1536)):
    c = CASES[case]
    head = HeadCls(in_channels=list(chans), inference_steps=c["T"], num_train_timesteps=1000, depth_feature_dim=16, loss_cfgs=[]).eval()
    fsd = {k: v for k, v in synth.make_fpn_state_dict(c["fseed"], in_channels=chans).items() if not k.startswith("convup_fp")}
    _load(head, synth.make_state_dict(c["wseed"], "swin", c["decoder_gain"], c["decoder_log_scale"]), fsd,
          synth.make_hahi_state_dict(c["hseed"], chans))
    B, H, W = c["B"], c["H"], c["W"]
    fp = [torch.from_numpy(f) for f in synth.make_backbone_features(c["iseed"], B, H // 2, W // 2, in_channels=chans)]   # strides 4..32
    gt = torch.from_numpy(synth.make_gt_depth(c["iseed"] + 1, B, H, W))
    h, w = synth.latent_hw(H, W)
    inp = synth.make_inputs(c["iseed"] + 2, B, h, w, (fp[0].shape[2], fp[0].shape[3]))
    with torch.no_grad():
        neck = head.hahineck(fp)
    with torch.no_grad(), inject_rng([inp["x_T"], inp["noise"]], [inp["timesteps"]]):
        o = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=False)
    assert set(o) == set(CASES["head_res"]["output_keys"])
    out = {"pred": t2n(o["pred"]), "pred_init": t2n(o["pred_init"]), "ddim_loss": t2n(o["ddim_loss"]).reshape(1),
           "state_keys": np.array(sorted(k for k in head.state_dict() if not k.endswith("num_batches_tracked")))}
    for i, n in enumerate(neck):
        out[f"neck{i}_ch0_2"] = t2n(n)[:, :2]
        out[f"neck{i}_sum"] = np.array([float(n.double().sum()), float(n.double().abs().max())])
    print(case + ": pred range", float(o["pred"].min()), float(o["pred"].max()), "ddim_loss", float(o["ddim_loss"]),
          "neck max", [float(n.abs().max()) for n in neck])
    return out


def gen_head_res_vis(VisCls):
    c = CASES["head_res_vis"]
    head = VisCls(in_channels=[64, 128, 256, 512], inference_steps=c["T"], num_train_timesteps=1000, depth_feature_dim=16, loss_cfgs=[]).eval()
    _load(head, synth.make_state_dict(c["wseed"], "res", c["decoder_gain"], c["decoder_log_scale"]), synth.make_fpn_state_dict(c["fseed"]))
    B, H, W = c["B"], c["H"], c["W"]
    fp = [torch.from_numpy(f) for f in synth.make_backbone_features(c["iseed"], B, H, W)]
    gt = torch.from_numpy(synth.make_gt_depth(c["iseed"] + 1, B, H, W))
    h, w = synth.latent_hw(H, W)
    inp = synth.make_inputs(c["iseed"] + 2, B, h, w)
    with torch.no_grad(), inject_rng([inp["x_T"], inp["noise"]], [inp["timesteps"]]):
        o = head(fp, gt, gt > 0, gt_depth_map=gt, return_loss=False)
    assert len(o["pred_inter"]) == c["T"]
    print("head_res_vis: pred max", float(o["pred"].max()), "first/last inter max", float(o["pred_inter"][0].max()), float(o["pred_inter"][-1].max()))
    return {"pred": t2n(o["pred"]), "pred_inter": np.stack([t2n(p) for p in o["pred_inter"]]), "ddim_loss": t2n(o["ddim_loss"]).reshape(1)}


def main():
    _, MpvitCls, HeadCls, VisCls = load_hahi_reference()
    np.savez_compressed(os.path.join(HERE, "head_swin_hahi.npz"), **gen_head_swin_hahi(HeadCls))
    np.savez_compressed(os.path.join(HERE, "head_mpvit_hahi.npz"), **gen_head_swin_hahi(MpvitCls, "head_mpvit_hahi", (128, 216, 288, 288)))
    np.savez_compressed(os.path.join(HERE, "head_res_vis.npz"), **gen_head_res_vis(VisCls))


if __name__ == "__main__":
    main()
