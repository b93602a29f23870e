"""TEST INFRASTRUCTURE ONLY -- fp64 NumPy restatement of the multi-scale deformable attention of the reference's HAHI neck
(src/model/necks/hahi.py:10,108-118 builds it, :211-223 and :235-247 call it) and of the positional encoding next to it (:105-106).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the product (diffusiondepth_amd/necks.py,
csrc/dd_msda.hip) never does.

**PARITY UNPINNED.**  The algorithm lives in a third-party dependency that is absent from /root/reference and from this image:
``mmcv-full`` (requirements.txt:84 pins 1.3.13; README.md:78 names 1.6.2) -- ``mmcv.ops.multi_scale_deform_attn.MultiScaleDeformableAttention`` with
its native operator ``ms_deform_attn_forward / _backward`` and ``mmcv.cnn.bricks.transformer.build_positional_encoding`` ->
``SinePositionalEncoding``.  What follows restates the PUBLISHED algorithm (Deformable DETR, Zhu et al., ICLR 2021, eq. 3 and its
released operator, which mmcv vendors) from its definition:

    out[b, q, m, :] = sum_l sum_p  A[b, q, m, l, p] * bilinear( V_l[b, :, m, :],  x = loc_x * W_l - 0.5,  y = loc_y * H_l - 0.5 )

with zero padding outside the level's map (== torch.nn.functional.grid_sample(mode="bilinear", padding_mode="zeros", align_corners=False) on
the grid 2 * loc - 1, which is the formulation ``ms_deform_attn_core_grid_sample`` below evaluates INDEPENDENTLY as a cross-check), and the module
around it: value projection, sampling offsets and attention logits as linear maps of (query + query_pos), softmax over the L * P samples of a
head, offsets normalised by (W_l, H_l), output projection, residual.  No golden vector can pin it: the reference's own tests hold none, mmcv
cannot be imported here, and the reference itself never executes it -- every DiffusionDepth head builds the neck with
``cross_att=False, self_att=False`` (ddim_depth_estimate_res_swin_addHAHI.py:54-56, ..._mpvit_HAHI.py:51-53), and with attention ON the neck as
those heads feed it (four inputs = three transformer levels, hahi.py:176,182) cannot run: the modules are built with num_levels = 4 (hahi.py:109-118)
and ``sampling_offsets.view(.., 4, P, 2) / offset_normalizer[.., 3, .., 2]`` does not broadcast.  It IS consistent for FIVE inputs (one
convolutional + four transformer levels, DepthFormer's own use of this neck); parity is anchored on the call sites above for that configuration.
"""
from __future__ import annotations

import numpy as np


def ms_deform_attn_core(value, spatial_shapes, sampling_locations, attention_weights, level_start_index=None):
    """The operator (ms_deform_attn_forward), directly from its definition.
    value (B, K, M, D); spatial_shapes [(H_l, W_l)]; sampling_locations (B, Q, M, L, P, 2) as (x, y) in [0, 1]; attention_weights (B, Q, M, L, P)
    -> (B, Q, M * D)."""
    value = np.asarray(value, np.float64)
    loc = np.asarray(sampling_locations, np.float64)
    attn = np.asarray(attention_weights, np.float64)
    B, K, M, D = value.shape
    _, Q, _, L, P, _ = loc.shape
    shapes = [(int(h), int(w)) for h, w in np.asarray(spatial_shapes).reshape(-1, 2)]
    assert len(shapes) == L
    if level_start_index is None:
        level_start_index = np.concatenate([[0], np.cumsum([h * w for h, w in shapes])[:-1]])
    out = np.zeros((B, Q, M, D))
    bi = np.arange(B)[:, None, None, None]
    mi = np.arange(M)[None, None, :, None]
    for l, (H, W) in enumerate(shapes):
        s0 = int(level_start_index[l])
        v = value[:, s0:s0 + H * W].reshape(B, H, W, M, D)
        x = loc[:, :, :, l, :, 0] * W - 0.5                      # (B, Q, M, P)
        y = loc[:, :, :, l, :, 1] * H - 0.5
        inside = (y > -1) & (x > -1) & (y < H) & (x < W)
        y0, x0 = np.floor(y).astype(np.int64), np.floor(x).astype(np.int64)
        ly, lx = y - y0, x - x0
        acc = np.zeros((B, Q, M, P, D))
        for dy, dx, wgt in ((0, 0, (1 - ly) * (1 - lx)), (0, 1, (1 - ly) * lx), (1, 0, ly * (1 - lx)), (1, 1, ly * lx)):
            yy, xx = y0 + dy, x0 + dx
            ok = inside & (yy >= 0) & (yy <= H - 1) & (xx >= 0) & (xx <= W - 1)
            g = v[bi, np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1), mi]          # (B, Q, M, P, D)
            acc += (wgt * ok)[..., None] * g
        out += (attn[:, :, :, l, :, None] * acc).sum(axis=3)
    return out.reshape(B, Q, M * D)


def ms_deform_attn_core_grid_sample(value, spatial_shapes, sampling_locations, attention_weights):
    """The same operator through torch's grid_sample (fp64): the formulation mmcv ships as its pure-PyTorch fallback, used here as an independent
    second evaluation (and, under autograd, as the gradient reference of the operator's backward)."""
    import torch
    import torch.nn.functional as F
    value = torch.as_tensor(value, dtype=torch.float64) if not isinstance(value, torch.Tensor) else value
    loc = torch.as_tensor(sampling_locations, dtype=torch.float64) if not isinstance(sampling_locations, torch.Tensor) else sampling_locations
    attn = torch.as_tensor(attention_weights, dtype=torch.float64) if not isinstance(attention_weights, torch.Tensor) else attention_weights
    B, K, M, D = value.shape
    _, Q, _, L, P, _ = loc.shape
    shapes = [(int(h), int(w)) for h, w in np.asarray(spatial_shapes).reshape(-1, 2)]
    grids = 2 * loc - 1
    start, sampled = 0, []
    for l, (H, W) in enumerate(shapes):
        v = value[:, start:start + H * W].flatten(2).transpose(1, 2).reshape(B * M, D, H, W)
        start += H * W
        g = grids[:, :, :, l].transpose(1, 2).flatten(0, 1)                        # (B * M, Q, P, 2)
        sampled.append(F.grid_sample(v, g, mode="bilinear", padding_mode="zeros", align_corners=False))      # (B * M, D, Q, P)
    a = attn.transpose(1, 2).reshape(B * M, 1, Q, L * P)
    out = (torch.stack(sampled, dim=-2).flatten(-2) * a).sum(-1).view(B, M * D, Q)
    return out.transpose(1, 2).contiguous()


def _softmax(x, axis=-1):
    x = x - x.max(axis=axis, keepdims=True)
    e = np.exp(x)
    return e / e.sum(axis=axis, keepdims=True)


def msda_module_forward(p, query, value=None, identity=None, query_pos=None, reference_points=None, spatial_shapes=None, level_start_index=None,
                        num_heads=8, num_levels=4, num_points=8):
    """MultiScaleDeformableAttention.forward with batch_first=True in eval mode (dropout = identity), as the neck calls it (hahi.py:212-221,
    236-245).  p: {"sampling_offsets.weight", ".bias", "attention_weights.*", "value_proj.*", "output_proj.*"} (torch Linear layout: y = x W^T + b)."""
    f = lambda a: np.asarray(a, np.float64)
    lin = lambda name, x: x @ f(p[name + ".weight"]).T + f(p[name + ".bias"])
    query = f(query)
    if value is None:
        value = query
    if identity is None:
        identity = query
    if query_pos is not None:
        query = query + f(query_pos)
    value = f(value)
    B, Q, E = query.shape
    K = value.shape[1]
    shapes = np.asarray(spatial_shapes).reshape(-1, 2)
    assert int((shapes[:, 0] * shapes[:, 1]).sum()) == K
    v = lin("value_proj", value).reshape(B, K, num_heads, E // num_heads)
    off = lin("sampling_offsets", query).reshape(B, Q, num_heads, num_levels, num_points, 2)
    aw = _softmax(lin("attention_weights", query).reshape(B, Q, num_heads, num_levels * num_points)).reshape(B, Q, num_heads, num_levels, num_points)
    ref = f(reference_points)
    assert ref.shape[-1] == 2
    normalizer = np.stack([shapes[:, 1], shapes[:, 0]], -1).astype(np.float64)          # (L', 2) as (W, H)
    loc = ref[:, :, None, :, None, :] + off / normalizer[None, None, None, :, None, :]    # num_levels != L' does not broadcast: the reference's dead end
    out = ms_deform_attn_core(v, shapes, loc, aw, level_start_index)
    return lin("output_proj", out) + f(identity)


def sine_positional_encoding(mask, num_feats=256, temperature=10000, normalize=False, scale=2 * np.pi, eps=1e-6, offset=0.0):
    """SinePositionalEncoding.forward(mask) (mask (B, H, W) bool, True = padded) -> (B, 2 * num_feats, H, W): the DETR encoding the neck builds
    from dict(type='SinePositionalEncoding', num_feats=256) (…swin_addHAHI.py:55; hahi.py:105-106,189,231).  mmcv computes it in fp32."""
    not_mask = 1 - np.asarray(mask).astype(np.int64)
    y_embed = np.cumsum(not_mask, axis=1).astype(np.float64)
    x_embed = np.cumsum(not_mask, axis=2).astype(np.float64)
    if normalize:
        y_embed = (y_embed + offset) / (y_embed[:, -1:, :] + eps) * scale
        x_embed = (x_embed + offset) / (x_embed[:, :, -1:] + eps) * scale
    dim_t = np.arange(num_feats, dtype=np.float64)
    dim_t = float(temperature) ** (2 * (dim_t // 2) / num_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    B, H, W = not_mask.shape
    pos_x = np.stack((np.sin(pos_x[:, :, :, 0::2]), np.cos(pos_x[:, :, :, 1::2])), axis=4).reshape(B, H, W, -1)
    pos_y = np.stack((np.sin(pos_y[:, :, :, 0::2]), np.cos(pos_y[:, :, :, 1::2])), axis=4).reshape(B, H, W, -1)
    return np.concatenate((pos_y, pos_x), axis=3).transpose(0, 3, 1, 2)


def reference_points_of_levels(spatial_shapes, valid_ratios):
    """HAHIHeteroNeck.get_reference_points (hahi.py:152-166): centres of every level's cells, normalised, per level -> (B, K, L, 2)."""
    vr = np.asarray(valid_ratios, np.float64)
    pts = []
    for lvl, (H, W) in enumerate(np.asarray(spatial_shapes).reshape(-1, 2)):
        ry, rx = np.meshgrid(np.linspace(0.5, H - 0.5, int(H)), np.linspace(0.5, W - 0.5, int(W)), indexing="ij")
        ry = ry.reshape(-1)[None] / (vr[:, None, lvl, 1] * H)
        rx = rx.reshape(-1)[None] / (vr[:, None, lvl, 0] * W)
        pts.append(np.stack((rx, ry), -1))
    ref = np.concatenate(pts, 1)
    return ref[:, :, None] * vr[:, None]
