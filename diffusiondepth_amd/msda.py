"""Multi-scale deformable attention over the C ABI of include/ddepth_msda.h: the operator under the reference neck's two
``mmcv.ops.MultiScaleDeformableAttention`` modules (src/model/necks/hahi.py:10,108-118,211-247) and the module itself.

Mirrored interfaces (mmcv-full, un-vendored: requirements.txt:84; same names, argument order and error behaviour):
  MultiScaleDeformableAttnFunction.apply(value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step)
  multi_scale_deformable_attn_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights)      -- the eager path for tensors NOT on a HIP device
  MultiScaleDeformableAttention(embed_dims, num_heads, num_levels, num_points, im2col_step, dropout, batch_first)
The arithmetic of the operator on a HIP device runs in libddepth_hip.so (csrc/dd_msda.hip), forward and backward; the linear projections
around it are torch.nn.Linear (rocBLAS GEMMs).  Tensors on the CPU take ``multi_scale_deformable_attn_pytorch`` -- the module's own eager
form (grid_sample), never the test oracle -- exactly as mmcv falls back when its extension has no CUDA tensor to work on.
"""
from __future__ import annotations

import ctypes
import math
import warnings

import torch
import torch.nn.functional as F
from torch import nn
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from . import backend

# every symbol include/ddepth_msda.h declares (checked by tests/test_msda_cpu.py)
ABI_SYMBOLS = ["dd_msda_last_error", "dd_msda_forward", "dd_msda_backward"]

_bound = None


def _lib():
    global _bound
    if _bound is None:
        lib = backend.load_library()
        c_int, c_vp = ctypes.c_int, ctypes.c_void_p
        lib.dd_msda_last_error.restype, lib.dd_msda_last_error.argtypes = ctypes.c_char_p, []
        lib.dd_msda_forward.restype, lib.dd_msda_forward.argtypes = c_int, [c_vp] * 6 + [c_int] * 8 + [c_vp]
        lib.dd_msda_backward.restype, lib.dd_msda_backward.argtypes = c_int, [c_vp] * 9 + [c_int] * 8 + [c_vp]
        _bound = lib
    return _bound


def _ck(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed ({rc}): {_lib().dd_msda_last_error().decode()}")


def _stream(t):
    return ctypes.c_void_p(int(torch.cuda.current_stream(t.device).cuda_stream))


def _dev(t, name, dtype):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be on a HIP device (the operator has no CPU path; CPU tensors take multi_scale_deformable_attn_pytorch)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    return t.contiguous()


def _shapes(value, spatial_shapes, level_start_index, loc, attn):
    if value.dim() != 4 or loc.dim() != 6 or attn.dim() != 5:
        raise RuntimeError("value must be (B, num_keys, heads, dims), sampling_locations (B, Q, heads, levels, points, 2), attention_weights (B, Q, heads, levels, points)")
    B, K, M, D = value.shape
    _, Q, M2, L, P, two = loc.shape
    if two != 2 or M2 != M or loc.shape[0] != B or tuple(attn.shape) != (B, Q, M, L, P):
        raise RuntimeError(f"inconsistent shapes: value {tuple(value.shape)}, sampling_locations {tuple(loc.shape)}, attention_weights {tuple(attn.shape)}")
    if tuple(spatial_shapes.shape) != (L, 2) or tuple(level_start_index.shape) != (L,):
        raise RuntimeError(f"spatial_shapes must be ({L}, 2) and level_start_index ({L},)")
    return B, K, M, D, L, Q, P


class MultiScaleDeformableAttnFunction(Function):
    """mmcv.ops.multi_scale_deform_attn.MultiScaleDeformableAttnFunction: forward -> dd_msda_forward, backward -> dd_msda_backward."""

    @staticmethod
    def forward(ctx, value, value_spatial_shapes, value_level_start_index, sampling_locations, attention_weights, im2col_step):
        value, loc, attn = (_dev(t, n, torch.float32) for t, n in ((value, "value"), (sampling_locations, "sampling_locations"),
                                                                   (attention_weights, "attention_weights")))
        shapes = _dev(value_spatial_shapes, "value_spatial_shapes", torch.int64)
        starts = _dev(value_level_start_index, "value_level_start_index", torch.int64)
        B, K, M, D, L, Q, P = _shapes(value, shapes, starts, loc, attn)
        ctx.im2col_step = int(im2col_step)
        out = torch.empty((B, Q, M * D), device=value.device, dtype=torch.float32)
        with torch.cuda.device(value.device):
            _ck(_lib().dd_msda_forward(value.data_ptr(), shapes.data_ptr(), starts.data_ptr(), loc.data_ptr(), attn.data_ptr(), out.data_ptr(),
                                       B, K, M, D, L, Q, P, ctx.im2col_step, _stream(value)), "dd_msda_forward")
        ctx.save_for_backward(value, shapes, starts, loc, attn)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        value, shapes, starts, loc, attn = ctx.saved_tensors
        B, K, M, D, L, Q, P = _shapes(value, shapes, starts, loc, attn)
        go = _dev(grad_output, "grad_output", torch.float32)
        need = ctx.needs_input_grad
        gv = torch.empty_like(value) if need[0] else None
        gl = torch.empty_like(loc) if need[3] else None
        ga = torch.empty_like(attn) if need[4] else None
        with torch.cuda.device(value.device):
            _ck(_lib().dd_msda_backward(value.data_ptr(), shapes.data_ptr(), starts.data_ptr(), loc.data_ptr(), attn.data_ptr(), go.data_ptr(),
                                        gv.data_ptr() if gv is not None else None, gl.data_ptr() if gl is not None else None,
                                        ga.data_ptr() if ga is not None else None, B, K, M, D, L, Q, P, ctx.im2col_step, _stream(value)),
                "dd_msda_backward")
        return gv, None, None, gl, ga, None


def multi_scale_deformable_attn_pytorch(value, value_spatial_shapes, sampling_locations, attention_weights):
    """mmcv's pure-PyTorch form of the operator (grid_sample per level): what tensors that are not on a HIP device run."""
    bs, _, num_heads, embed_dims = value.shape
    _, num_queries, num_heads, num_levels, num_points, _ = sampling_locations.shape
    value_list = value.split([int(H_) * int(W_) for H_, W_ in value_spatial_shapes], dim=1)
    sampling_grids = 2 * sampling_locations - 1
    sampling_value_list = []
    for level, (H_, W_) in enumerate(value_spatial_shapes):
        value_l_ = value_list[level].flatten(2).transpose(1, 2).reshape(bs * num_heads, embed_dims, int(H_), int(W_))
        sampling_grid_l_ = sampling_grids[:, :, :, level].transpose(1, 2).flatten(0, 1)
        sampling_value_list.append(F.grid_sample(value_l_, sampling_grid_l_, mode="bilinear", padding_mode="zeros", align_corners=False))
    attention_weights = attention_weights.transpose(1, 2).reshape(bs * num_heads, 1, num_queries, num_levels * num_points)
    output = (torch.stack(sampling_value_list, dim=-2).flatten(-2) * attention_weights).sum(-1).view(bs, num_heads * embed_dims, num_queries)
    return output.transpose(1, 2).contiguous()


class MultiScaleDeformableAttention(nn.Module):
    """mmcv.ops.MultiScaleDeformableAttention: same constructor keywords, parameter tree (``sampling_offsets``, ``attention_weights``,
    ``value_proj``, ``output_proj``), init_weights and forward as the neck calls it (hahi.py:108-118,212-221,236-245)."""

    def __init__(self, embed_dims=256, num_heads=8, num_levels=4, num_points=4, im2col_step=64, dropout=0.1, batch_first=False,
                 norm_cfg=None, init_cfg=None):
        super().__init__()
        if embed_dims % num_heads != 0:
            raise ValueError(f"embed_dims must be divisible by num_heads, but got {embed_dims} and {num_heads}")
        dim_per_head = embed_dims // num_heads
        if dim_per_head & (dim_per_head - 1):
            warnings.warn("You'd better set embed_dims in MultiScaleDeformAttention to make the dimension of each attention head a power of 2 "
                          "which is more efficient in our CUDA implementation.")
        self.norm_cfg = norm_cfg
        self.dropout = nn.Dropout(dropout)
        self.batch_first = batch_first
        self.im2col_step = im2col_step
        self.embed_dims, self.num_levels, self.num_heads, self.num_points = embed_dims, num_levels, num_heads, num_points
        self.sampling_offsets = nn.Linear(embed_dims, num_heads * num_levels * num_points * 2)
        self.attention_weights = nn.Linear(embed_dims, num_heads * num_levels * num_points)
        self.value_proj = nn.Linear(embed_dims, embed_dims)
        self.output_proj = nn.Linear(embed_dims, embed_dims)
        self.init_weights()

    def init_weights(self):
        """Zero offsets weights with the heads' directions in the bias, uniform attention, Xavier projections (Deformable DETR's initialisation)."""
        nn.init.constant_(self.sampling_offsets.weight, 0.)
        thetas = torch.arange(self.num_heads, dtype=torch.float32) * (2.0 * math.pi / self.num_heads)
        grid_init = torch.stack([thetas.cos(), thetas.sin()], -1)
        grid_init = (grid_init / grid_init.abs().max(-1, keepdim=True)[0]).view(self.num_heads, 1, 1, 2).repeat(1, self.num_levels, self.num_points, 1)
        for i in range(self.num_points):
            grid_init[:, :, i, :] *= i + 1
        with torch.no_grad():
            self.sampling_offsets.bias.copy_(grid_init.view(-1))
        nn.init.constant_(self.attention_weights.weight, 0.)
        nn.init.constant_(self.attention_weights.bias, 0.)
        nn.init.xavier_uniform_(self.value_proj.weight)
        nn.init.constant_(self.value_proj.bias, 0.)
        nn.init.xavier_uniform_(self.output_proj.weight)
        nn.init.constant_(self.output_proj.bias, 0.)

    def forward(self, query, key=None, value=None, identity=None, query_pos=None, key_padding_mask=None, reference_points=None,
                spatial_shapes=None, level_start_index=None, **kwargs):
        if value is None:
            value = query
        if identity is None:
            identity = query
        if query_pos is not None:
            query = query + query_pos
        if not self.batch_first:
            query, value = query.permute(1, 0, 2), value.permute(1, 0, 2)
        bs, num_query, _ = query.shape
        bs, num_value, _ = value.shape
        assert int((spatial_shapes[:, 0] * spatial_shapes[:, 1]).sum()) == num_value
        value = self.value_proj(value)
        if key_padding_mask is not None:
            value = value.masked_fill(key_padding_mask[..., None], 0.0)
        value = value.view(bs, num_value, self.num_heads, -1)
        sampling_offsets = self.sampling_offsets(query).view(bs, num_query, self.num_heads, self.num_levels, self.num_points, 2)
        attention_weights = self.attention_weights(query).view(bs, num_query, self.num_heads, self.num_levels * self.num_points)
        attention_weights = attention_weights.softmax(-1).view(bs, num_query, self.num_heads, self.num_levels, self.num_points)
        if reference_points.shape[-1] == 2:
            offset_normalizer = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1)
            # (num_levels against the rows of spatial_shapes: a mismatch -- the reference neck's 4 against three transformer levels -- fails HERE, as in mmcv)
            sampling_locations = reference_points[:, :, None, :, None, :] + sampling_offsets / offset_normalizer[None, None, None, :, None, :]
        elif reference_points.shape[-1] == 4:
            sampling_locations = reference_points[:, :, None, :, None, :2] + sampling_offsets / self.num_points * reference_points[:, :, None, :, None, 2:] * 0.5
        else:
            raise ValueError(f"Last dim of reference_points must be 2 or 4, but get {reference_points.shape[-1]} instead.")
        if value.is_cuda:
            output = MultiScaleDeformableAttnFunction.apply(value.float(), spatial_shapes, level_start_index, sampling_locations.float(),
                                                            attention_weights.float(), self.im2col_step).to(query.dtype)
        else:
            output = multi_scale_deformable_attn_pytorch(value, spatial_shapes, sampling_locations, attention_weights)
        output = self.output_proj(output)
        if not self.batch_first:
            output = output.permute(1, 0, 2)
        return self.dropout(output) + identity


class SinePositionalEncoding(nn.Module):
    """mmcv.cnn.bricks.transformer's 'SinePositionalEncoding' (the neck builds it from dict(type='SinePositionalEncoding', num_feats=256),
    …swin_addHAHI.py:55, hahi.py:105-106): forward(mask (B, H, W) bool) -> (B, 2 * num_feats, H, W).  Parameter-free."""

    def __init__(self, num_feats, temperature=10000, normalize=False, scale=2 * math.pi, eps=1e-6, offset=0., init_cfg=None):
        super().__init__()
        if normalize and not isinstance(scale, (float, int)):
            raise AssertionError(f"when normalize is set, scale should be provided and in float or int type, found {type(scale)}")
        self.num_feats, self.temperature, self.normalize, self.scale, self.eps, self.offset = num_feats, temperature, normalize, scale, eps, offset

    def forward(self, mask):
        mask = mask.to(torch.int)
        not_mask = 1 - mask
        y_embed = not_mask.cumsum(1, dtype=torch.float32)
        x_embed = not_mask.cumsum(2, dtype=torch.float32)
        if self.normalize:
            y_embed = (y_embed + self.offset) / (y_embed[:, -1:, :] + self.eps) * self.scale
            x_embed = (x_embed + self.offset) / (x_embed[:, :, -1:] + self.eps) * self.scale
        dim_t = torch.arange(self.num_feats, dtype=torch.float32, device=mask.device)
        dim_t = self.temperature ** (2 * (dim_t // 2) / self.num_feats)
        pos_x = x_embed[:, :, :, None] / dim_t
        pos_y = y_embed[:, :, :, None] / dim_t
        B, H, W = mask.size()
        pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).view(B, H, W, -1)
        pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).view(B, H, W, -1)
        return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


def build_positional_encoding(cfg):
    """mmcv's registry lookup for the one type the reference configures."""
    cfg = dict(cfg)
    kind = cfg.pop("type")
    if kind != "SinePositionalEncoding":
        raise KeyError(f"{kind} is not in the positional encoding registry of this package (the reference heads configure 'SinePositionalEncoding')")
    return SinePositionalEncoding(**cfg)
