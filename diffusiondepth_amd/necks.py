"""HAHIHeteroNeck as the DiffusionDepth heads use it (reference src/model/necks/hahi.py:15-275): same constructor, same parameter
tree (state_dicts of the reference load with strict=True), same forward(inputs) -> list of 4 maps.

The neck stays in PyTorch-ROCm (north star: backbone-side feature extraction is not part of the HIP hot path):
it runs once per image in front of the condition FPN.  Every shipped head builds it with ``cross_att=False, self_att=False``
(src/model/head/ddim_depth_estimate_res_swin_addHAHI.py:54-56, ..._mpvit_HAHI.py:51-53), for which the forward reduces to
    l_i   = lateral_convs[i](x_i)                                  1x1 conv + BN + ReLU                     (hahi.py:170-173)
    out_0 = conv_fusion(cat[conv_proj(l_0), l_0])                  1x1 -> 512, then 3x3 (C_0+512 -> C_0)    (hahi.py:226-249)
    out_i = trans_fusion[i-1](cat[l_i, trans_proj[i-1](l_i)])      1x1 -> 512, then 3x3 (C_i+512 -> C_i)    (hahi.py:196-197,252-272)
Attention (round 6).  ``cross_att`` / ``self_att`` switch on the two MultiScaleDeformableAttention modules
(hahi.py:108-118): the hierarchical self attention over the flattened transformer levels (:176-223) and the cross attention of the
convolutional level onto them (:225-247).  Module and operator are diffusiondepth_amd.msda (mmcv's interfaces; the operator in HIP,
csrc/dd_msda.hip, forward and backward).  The modules are built with num_levels = 4 as in the reference (:109-118), so -- exactly as there --
attention runs for FIVE inputs (one convolutional + four transformer levels: DepthFormer's use of this neck) and fails for the four inputs
the DiffusionDepth heads feed it (three transformer levels: mmcv's offset normalisation does not broadcast 4 against 3), which is why every
head passes ``cross_att=False, self_att=False``.  No reference-side run of this path exists (mmcv-full is un-vendored and absent): parity
is against oracle/msda_oracle.py, whose header says "parity unpinned".
"""
from __future__ import annotations

import torch
from torch import nn

from .msda import MultiScaleDeformableAttention, build_positional_encoding


class _ConvModule(nn.Module):
    """mmcv.cnn.ConvModule(conv + BN + ReLU) parameter layout: ``conv`` (bias-free because a norm follows), ``bn``, ``activate``."""

    def __init__(self, cin, cout, kernel_size, padding=0, stride=1):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, kernel_size, stride, padding, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        self.activate = nn.ReLU(inplace=True)

    def forward(self, x):
        return self.activate(self.bn(self.conv(x)))


class HAHIHeteroNeck(nn.Module):
    def __init__(self, in_channels, out_channels, embedding_dim, positional_encoding=None, scales=(1, 1, 1, 1), norm_cfg=None,
                 act_cfg=None, cross_att=True, self_att=True, num_points=8):
        super().__init__()
        assert isinstance(in_channels, list)
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
        num_feature_levels = 4                                                                               # hahi.py:103
        # parameter-free; the heads' dict(type='SinePositionalEncoding', num_feats=256) (None: attention off, nothing to encode)
        self.trans_positional_encoding = build_positional_encoding(positional_encoding) if positional_encoding is not None else None
        self.conv_positional_encoding = build_positional_encoding(positional_encoding) if positional_encoding is not None else None
        self.reference_points = nn.Linear(embedding_dim, 2)
        self.level_embed = nn.Parameter(torch.zeros(num_feature_levels, embedding_dim))
        self.multi_att = MultiScaleDeformableAttention(embed_dims=embedding_dim, num_levels=4, num_heads=8, num_points=num_points, batch_first=True)
        self.self_attn = MultiScaleDeformableAttention(embed_dims=embedding_dim, num_levels=4, num_heads=8, num_points=num_points, batch_first=True)

    def init_weights(self):
        """hahi.py:120-134."""
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)
        nn.init.xavier_uniform_(self.reference_points.weight.data, gain=1.0)
        nn.init.constant_(self.reference_points.bias.data, 0.)
        nn.init.normal_(self.level_embed)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight, gain=1)            # mmcv xavier_init(distribution='uniform'); the neck's convs carry no bias
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            if isinstance(m, MultiScaleDeformableAttention):
                m.init_weights()

    @staticmethod
    def get_valid_ratio(mask):
        _, H, W = mask.shape
        valid_H = torch.sum(~mask[:, :, 0], 1)
        valid_W = torch.sum(~mask[:, 0, :], 1)
        return torch.stack([valid_W.float() / W, valid_H.float() / H], -1)

    @staticmethod
    def get_reference_points(spatial_shapes, valid_ratios, device):
        """Cell centres of every level, normalised (hahi.py:152-166) -> (B, sum H_l W_l, levels, 2)."""
        pts = []
        for lvl, (H_, W_) in enumerate(spatial_shapes):
            H_, W_ = int(H_), int(W_)
            ref_y, ref_x = torch.meshgrid(torch.linspace(0.5, H_ - 0.5, H_, dtype=torch.float32, device=device),
                                          torch.linspace(0.5, W_ - 0.5, W_, dtype=torch.float32, device=device), indexing="ij")
            ref_y = ref_y.reshape(-1)[None] / (valid_ratios[:, None, lvl, 1] * H_)
            ref_x = ref_x.reshape(-1)[None] / (valid_ratios[:, None, lvl, 0] * W_)
            pts.append(torch.stack((ref_x, ref_y), -1))
        reference_points = torch.cat(pts, 1)
        return reference_points[:, :, None] * valid_ratios[:, None]

    def forward(self, inputs):
        assert len(inputs) == len(self.in_channels)
        feats = [conv(inputs[i]) for i, conv in enumerate(self.lateral_convs)]                              # hahi.py:170-173
        if not (self.cross_att or self.self_att):
            # what every DiffusionDepth head runs: no attention, nothing flattened
            outs = [self.conv_fusion(torch.cat([self.conv_proj(feats[0]), feats[0]], dim=1))]               # :226-249 (query = conv_skip)
            for i, f in enumerate(feats[1:]):
                outs.append(self._fuse_trans(i, torch.cat([f, self.trans_proj[i](f)], dim=1)))              # :196-197, :252-272
            return outs
        if self.trans_positional_encoding is None:
            raise ValueError("HAHIHeteroNeck with attention needs positional_encoding=dict(type='SinePositionalEncoding', num_feats=embedding_dim // 2)")
        feats_trans, feat_conv = feats[1:], feats[0]
        # HI: the transformer levels as one token sequence (hahi.py:176-209)
        masks, srcs, pos_embeds, shapes = [], [], [], []
        for i, f in enumerate(feats_trans):
            bs, _, h, w = f.shape
            shapes.append((h, w))
            mask = torch.zeros((bs, h, w), dtype=torch.bool, device=f.device)
            masks.append(mask)
            pos = self.trans_positional_encoding(mask).flatten(2).transpose(1, 2)
            pos_embeds.append(pos + self.level_embed[i].view(1, 1, -1))
            srcs.append(self.trans_proj[i](f).flatten(2).transpose(1, 2))
        src_flatten = torch.cat(srcs, 1)
        lvl_pos_embed_flatten = torch.cat(pos_embeds, 1)
        spatial_shapes = torch.as_tensor(shapes, dtype=torch.long, device=src_flatten.device)
        level_start_index = torch.cat((spatial_shapes.new_zeros((1,)), spatial_shapes.prod(1).cumsum(0)[:-1]))
        valid_ratios = torch.stack([self.get_valid_ratio(m) for m in masks], 1)
        if self.self_att:
            reference_points = self.get_reference_points(shapes, valid_ratios, device=src_flatten.device)
            src = self.self_attn(src_flatten, key=None, value=None, identity=None, query_pos=lvl_pos_embed_flatten, key_padding_mask=None,
                                 reference_points=reference_points, spatial_shapes=spatial_shapes, level_start_index=level_start_index)      # :211-221
        else:
            src = src_flatten
        # HA: the convolutional level queries the transformer tokens (hahi.py:225-249)
        conv_skip = self.conv_proj(feat_conv)
        bs, c, h, w = conv_skip.shape
        query = conv_skip.flatten(2).transpose(1, 2)
        if self.cross_att:
            query_mask = torch.zeros((bs, h, w), dtype=torch.bool, device=conv_skip.device)
            query_embed = self.conv_positional_encoding(query_mask).flatten(2).transpose(1, 2)
            reference_points = self.reference_points(query_embed).sigmoid()
            reference_points_input = reference_points[:, :, None] * valid_ratios[:, None]
            fused = self.multi_att(query, key=None, value=src, identity=None, query_pos=query_embed, key_padding_mask=None,
                                   reference_points=reference_points_input, spatial_shapes=spatial_shapes, level_start_index=level_start_index)   # :235-245
        else:
            fused = query
        outs = [self.conv_fusion(torch.cat([fused.permute(0, 2, 1).reshape(bs, c, h, w), feat_conv], dim=1))]
        # the tokens back to their maps, next to the lateral features, through the 3x3 fusion convs (hahi.py:252-272)
        start = 0
        for i, f in enumerate(feats_trans):
            bs, _, h, w = f.shape
            tok = src[:, start:start + h * w, :].permute(0, 2, 1).reshape(bs, self.embedding_dim, h, w)
            start += h * w
            outs.append(self._fuse_trans(i, torch.cat([f, tok], dim=1)))
        return outs

    def _fuse_trans(self, i, x):
        if self.scales[i] != 1:
            raise NotImplementedError("scales != 1 hits `from model.ops import resize` binding a module in the reference (hahi.py:12,268)")
        return self.trans_fusion[i](x)
