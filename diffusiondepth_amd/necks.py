"""HAHIHeteroNeck as the DiffusionDepth heads use it (reference src/model/necks/hahi.py:15-275): same constructor, same parameter
tree (state_dicts of the reference load with strict=True), same forward(inputs) -> list of 4 maps.

The neck stays in PyTorch-ROCm (north star: backbone-side feature extraction is not part of the HIP hot path):
it runs once per image in front of the condition FPN.  Every shipped head builds it with ``cross_att=False, self_att=False``
(src/model/head/ddim_depth_estimate_res_swin_addHAHI.py:54-56, ..._mpvit_HAHI.py:51-53), for which the forward reduces to
    l_i   = lateral_convs[i](x_i)                                  1x1 conv + BN + ReLU                     (hahi.py:170-173)
    out_0 = conv_fusion(cat[conv_proj(l_0), l_0])                  1x1 -> 512, then 3x3 (C_0+512 -> C_0)    (hahi.py:226-249)
    out_i = trans_fusion[i-1](cat[l_i, trans_proj[i-1](l_i)])      1x1 -> 512, then 3x3 (C_i+512 -> C_i)    (hahi.py:196-197,252-272)
The two MultiScaleDeformableAttention modules, ``reference_points`` and ``level_embed`` exist in the reference only as parameters
(constructed, never executed); they are kept here as parameter containers so checkpoints load.  Switching the attention ON is not
offered: the reference cannot run it either -- it builds the attention with num_levels=4 but feeds it the 3 transformer levels
``feats_projed[1:]`` (hahi.py:109-118,176,182), which mmcv's MultiScaleDeformableAttention rejects when it reshapes the sampling
offsets against the reference points.
"""
from __future__ import annotations

import torch
from torch import nn


class _ConvModule(nn.Module):
    """mmcv.cnn.ConvModule(conv + BN + ReLU) parameter layout: ``conv`` (bias-free because a norm follows), ``bn``, ``activate``."""

    def __init__(self, cin, cout, kernel_size, padding=0, stride=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size, stride, padding, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        self.activate = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.activate(self.bn(self.conv(x)))


class _MSDeformAttnParams(nn.Module):
    """Parameter container with the names / shapes of mmcv.ops.MultiScaleDeformableAttention(embed_dims, num_levels=4, num_heads=8,
    num_points) (un-vendored mmcv-full).  Never executed: see the module docstring."""

    def __init__(self, embed_dims, num_levels=4, num_heads=8, num_points=8):
        super().__init__()
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)

    def forward(self, *a, **k):
        raise NotImplementedError("HAHI attention is dead code in every DiffusionDepth head (cross_att=False, self_att=False) and cannot "
                                  "run in the reference as constructed (num_levels=4 vs 3 transformer levels)")


class HAHIHeteroNeck(nn.Module):
    def __init__(self, in_channels, out_channels, embedding_dim, positional_encoding=None, scales=(1, 1, 1, 1), norm_cfg=None,
                 act_cfg=None, cross_att=True, self_att=True, num_points=8):
        super().__init__()
        assert isinstance(in_channels, list)
        if cross_att or self_att:
            raise NotImplementedError("HAHIHeteroNeck(cross_att / self_att = True): not runnable in the reference either (see module docstring); "
                                      "every DiffusionDepth head passes False")
        if any(s != 1 for s in scales):
            raise NotImplementedError("scales != 1 hits `from model.ops import resize` binding a module in the reference (hahi.py:12,268)")
        self.cross_att, self.self_att = cross_att, self_att
        self.in_channels, self.out_channels = in_channels, out_channels
        self.scales = list(scales)
        self.num_outs = len(scales)
        self.embedding_dim = embedding_dim
        self.lateral_convs = nn.ModuleList(_ConvModule(ci, co, 1) for ci, co in zip(in_channels, out_channels))
        self.trans_proj = nn.ModuleList(_ConvModule(co, embedding_dim, 1) for co in out_channels[1:])
        self.trans_fusion = nn.ModuleList(_ConvModule(co + embedding_dim, co, 3, padding=1) for co in out_channels[1:])
        self.conv_proj = nn.Sequential(_ConvModule(in_channels[0], embedding_dim, 1))
        self.conv_fusion = nn.Sequential(_ConvModule(in_channels[0] + embedding_dim, out_channels[0], 3, padding=1))
        self.reference_points = nn.Linear(embedding_dim, 2)
        self.level_embed = nn.Parameter(torch.zeros(4, embedding_dim))
        self.multi_att = _MSDeformAttnParams(embedding_dim, 4, 8, num_points)
        self.self_attn = _MSDeformAttnParams(embedding_dim, 4, 8, num_points)

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        feats = [conv(inputs[i]) for i, conv in enumerate(self.lateral_convs)]                              # hahi.py:170-173
        outs = [self.conv_fusion(torch.cat([self.conv_proj(feats[0]), feats[0]], dim=1))]                   # :226-249 (query = conv_skip)
        for i, f in enumerate(feats[1:]):
            outs.append(self.trans_fusion[i](torch.cat([f, self.trans_proj[i](f)], dim=1)))                 # :196-197, :252-272
        return outs
