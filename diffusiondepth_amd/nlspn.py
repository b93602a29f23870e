"""Host-side mirror of the reference's ``NLSPN`` propagation module (src/model/nlspnmodel.py:22-207): same constructor,
parameter names / shapes (state_dicts load unchanged) and forward() contract.  The refinement's arithmetic runs in
libddepth_hip.so:

  * no autograd needed (inference; the BASELINE config "… + NLSPN refine" eval): two fused kernels, dd_nlspn_offset_affinity +
    dd_nlspn_propagate (prop_time launches), instead of the reference's 8 + 18 im2col/GEMM pairs;
  * autograd needed (training): torch tensor ops plus ``ModulatedDeformConvFunction`` (the HIP DCNv2 forward / backward of
    diffusiondepth_amd.dcn) once per confidence neighbour and once per iteration, as the reference's autograd graph has it.

``conv_offset_aff`` (ch_g -> 3*num, 3x3) is evaluated inside the affinity kernel for NLSPNModel's geometry (ch_g 8, 3x3, 3x3:
dd_nlspn_guided_offset_affinity) and is a torch convolution (MIOpen) otherwise and in training; the encoder-decoder producing guidance /
confidence / initial depth is PyTorch-ROCm (north star: "NLSPN-style refinement stay in PyTorch-ROCm" apart from this hot stage).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import dcn
from .dcn import ModulatedDeformConvFunction


class NLSPN(nn.Module):
    def __init__(self, args, ch_g, ch_f, k_g, k_f):
        """Same signature, attributes and parameter tree as the reference module (nlspnmodel.py:23-85): ``args`` carries prop_time,
        affinity, affinity_gamma, conf_prop, preserve_input, legacy (src/config.py:78-112)."""
        super().__init__()
        if ch_f != 1:
            raise AssertionError('only tested with ch_f == 1 but {}'.format(ch_f))            # the reference's assert (:30)
        for name, k in (("k_g", k_g), ("k_f", k_f)):
            if k % 2 != 1:
                raise AssertionError('only odd kernel is supported but {} = {}'.format(name, k))   # (:32-36)
        if args.affinity not in ('AS', 'ASS', 'TC', 'TGASS'):
            raise NotImplementedError(args.affinity)                                           # (:69-70)
        self.args = args
        self.prop_time, self.affinity = args.prop_time, args.affinity
        self.ch_g, self.ch_f, self.k_g, self.k_f = ch_g, ch_f, k_g, k_f
        self.num = k_f * k_f - 1                 # neighbours; the centre pixel has a fixed zero offset (:46-48)
        self.idx_ref = self.num // 2
        # guidance -> (2 offsets + 1 affinity) per neighbour; zero-initialised like the reference (:50-56)
        self.conv_offset_aff = nn.Conv2d(ch_g, 3 * self.num, kernel_size=k_g, stride=1, padding=(k_g - 1) // 2, bias=True)
        nn.init.zeros_(self.conv_offset_aff.weight)
        nn.init.zeros_(self.conv_offset_aff.bias)
        # affinity scale: num (TC, frozen), affinity_gamma * num (TGASS, trainable), 1 (AS / ASS, frozen)   (:58-68)
        scale = self.num if self.affinity == 'TC' else (args.affinity_gamma * self.num if self.affinity == 'TGASS' else 1.0)
        self.aff_scale_const = nn.Parameter(torch.full((1,), float(scale)), requires_grad=self.affinity == 'TGASS')
        # frozen all-ones "weight" / zero "bias" of the gathering operator, and the 1x1 weight of the confidence sampling (:73-80)
        self.w = nn.Parameter(torch.ones(ch_f, 1, k_f, k_f), requires_grad=False)
        self.b = nn.Parameter(torch.zeros(ch_f), requires_grad=False)
        self.w_conf = nn.Parameter(torch.ones(1, 1, 1, 1), requires_grad=False)
        self.stride, self.padding, self.dilation = 1, (k_f - 1) // 2, 1
        self.groups, self.deformable_groups, self.im2col_step = ch_f, 1, 64
        self.fuse_guidance_conv = True      # inference path: dd_nlspn_guided_offset_affinity where its geometry applies (A/B switch)

    # -- autograd path: same arithmetic as nlspnmodel.py:87-171, written on whole tensors ------------------------------------------
    def _neighbour_shift(self, like):
        """--legacy (nlspnmodel.py:126-134): neighbour m = (hh, ww) gets (hh - c, ww - c) added to its offsets, in place on the tensor
        that is later returned and propagated with; as a constant (2*(num+1),) vector here."""
        c = (self.k_f - 1) / 2
        sh = [(m // self.k_f - c, m % self.k_f - c) if m != self.idx_ref else (0.0, 0.0) for m in range(self.num + 1)]
        return torch.tensor(sh, dtype=like.dtype, device=like.device).view(1, -1, 1, 1)

    def _get_offset_affinity(self, guidance, confidence=None, rgb=None):
        B, _, H, W = guidance.shape
        num, ref = self.num, self.idx_ref
        raw = self.conv_offset_aff(guidance)                                   # (B, 3*num, H, W)
        # the first 2*num channels, pairwise, are the (row, column) offsets of the num neighbours (:91-95); the reference pixel gets (0, 0)
        pairs = raw[:, :2 * num].reshape(B, num, 2, H, W)
        offset = torch.cat([pairs[:, :ref], pairs.new_zeros(B, 1, 2, H, W), pairs[:, ref:]], dim=1).reshape(B, 2 * (num + 1), H, W)
        aff = raw[:, 2 * num:]
        if self.affinity == 'TC':
            aff = torch.tanh(aff) / self.aff_scale_const                       # :103-104
        elif self.affinity == 'TGASS':
            aff = torch.tanh(aff) / (self.aff_scale_const + 1e-8)              # :105-106
        if self.args.conf_prop:
            if self.args.legacy:
                offset = offset + self._neighbour_shift(offset)
            where = offset.detach()                                            # no gradient reaches the offsets through the confidence (:122)
            ones = where.new_ones(B, 1, H, W)
            sampled = [ModulatedDeformConvFunction.apply(confidence, where[:, 2 * m:2 * m + 2].contiguous(), ones, self.w_conf, self.b,
                                                         self.stride, 0, self.dilation, self.groups, self.deformable_groups, self.im2col_step)
                       for m in range(num + 1) if m != ref]                     # 1x1 sampling of the confidence at pixel + offset (:136-141)
            aff = aff * torch.cat(sampled, dim=1)                               # :143-144
        total = aff.abs().sum(dim=1, keepdim=True) + 1e-4                       # :147-148
        if self.affinity in ('ASS', 'TGASS'):
            total = total.clamp_min(1.0)                                        # :150-151 (values below 1 become the constant 1)
        if self.affinity in ('AS', 'ASS', 'TGASS'):
            aff = aff / total                                                   # :153-154
        centre = 1.0 - aff.sum(dim=1, keepdim=True)                             # :156-157
        return offset, torch.cat([aff[:, :ref], centre, aff[:, ref:]], dim=1)   # :159-161

    def _propagate_once(self, feat, offset, aff):
        return ModulatedDeformConvFunction.apply(feat.contiguous(), offset.contiguous(), aff.contiguous(), self.w, self.b, self.stride,
                                                 self.padding, self.dilation, self.groups, self.deformable_groups, self.im2col_step)

    # -- fused inference path ------------------------------------------------------------------------------------------
    def _needs_autograd(self, *tensors):
        if not torch.is_grad_enabled():
            return False
        return any(t is not None and t.requires_grad for t in tensors) or any(p.requires_grad for p in self.parameters())

    def forward(self, feat_init, guidance, confidence=None, feat_fix=None, rgb=None):
        assert self.ch_g == guidance.shape[1]
        assert self.ch_f == feat_init.shape[1]
        if self.args.conf_prop:
            assert confidence is not None
        dcn.require_hip(feat_init, "NLSPN")
        if self.args.preserve_input:
            assert feat_init.shape == feat_fix.shape
        if not self._needs_autograd(feat_init, guidance, confidence):
            conf = confidence if self.args.conf_prop else None
            if self.fuse_guidance_conv and dcn.guided_supported(self.ch_g, self.k_g, self.k_f):
                # conv_offset_aff evaluated inside the affinity kernel (NLSPNModel's geometry): no 24-plane intermediate
                offset, aff = dcn.nlspn_guided_offset_affinity(guidance, self.conv_offset_aff.weight, self.conv_offset_aff.bias, conf,
                                                               self.aff_scale_const, self.w_conf, self.b, self.k_g, self.k_f, self.affinity,
                                                               self.args.conf_prop, self.args.legacy)
            else:
                offset_aff = self.conv_offset_aff(guidance)          # torch / MIOpen
                offset, aff = dcn.nlspn_offset_affinity(offset_aff, conf, self.aff_scale_const, self.w_conf, self.b, self.k_f,
                                                        self.affinity, self.args.conf_prop, self.args.legacy)
            feats = dcn.nlspn_propagate(feat_init, offset, aff, feat_fix if self.args.preserve_input else None, self.w, self.b,
                                        self.k_f, self.prop_time, self.args.preserve_input)
            list_feat = list(feats.unbind(0))
            return list_feat[-1], list_feat, offset, aff, self.aff_scale_const.data
        # training: one DCNv2 operator call per neighbour / per iteration, differentiable end to end (nlspnmodel.py:176-207)
        offset, aff = self._get_offset_affinity(guidance, confidence if self.args.conf_prop else None, rgb)
        keep = (feat_fix > 0.0).any(dim=1, keepdim=True) if self.args.preserve_input else None      # mask_fix (:189-191)
        feat, list_feat = feat_init, []
        for _ in range(self.prop_time):
            if keep is not None:
                feat = torch.where(keep, feat_fix, feat)                       # (1 - mask_fix) * feat + mask_fix * feat_fix (:199-201)
            feat = self._propagate_once(feat, offset, aff)
            list_feat.append(feat)
        return feat, list_feat, offset, aff, self.aff_scale_const.data
