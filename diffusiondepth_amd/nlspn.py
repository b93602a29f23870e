"""Host-side mirror of the reference's ``NLSPN`` propagation module (src/model/nlspnmodel.py:22-207): same constructor,
parameter names / shapes (state_dicts load unchanged) and forward() contract.  The refinement's arithmetic runs in
libddepth_hip.so:

  * no autograd needed (inference; the BASELINE config "… + NLSPN refine" eval): two fused kernels, dd_nlspn_offset_affinity +
    dd_nlspn_propagate (prop_time launches), instead of the reference's 8 + 18 im2col/GEMM pairs;
  * autograd needed (training): the reference's own formulation -- torch ops plus ``ModulatedDeformConvFunction`` once per
    confidence neighbour and once per iteration -- with the HIP DCNv2 forward / backward of diffusiondepth_amd.dcn under it.

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
        super().__init__()
        assert ch_f == 1, 'only tested with ch_f == 1 but {}'.format(ch_f)                # nlspnmodel.py:30
        assert (k_g % 2) == 1, 'only odd kernel is supported but k_g = {}'.format(k_g)    # :32-33
        pad_g = int((k_g - 1) / 2)
        assert (k_f % 2) == 1, 'only odd kernel is supported but k_f = {}'.format(k_f)    # :35-36
        pad_f = int((k_f - 1) / 2)
        self.args = args
        self.prop_time = self.args.prop_time
        self.affinity = self.args.affinity
        self.ch_g, self.ch_f, self.k_g, self.k_f = ch_g, ch_f, k_g, k_f
        self.num = self.k_f * self.k_f - 1          # zero offset for the centre pixel (:46-48)
        self.idx_ref = self.num // 2
        if self.affinity in ['AS', 'ASS', 'TC', 'TGASS']:
            self.conv_offset_aff = nn.Conv2d(self.ch_g, 3 * self.num, kernel_size=self.k_g, stride=1, padding=pad_g, bias=True)
            self.conv_offset_aff.weight.data.zero_()                                      # :55-56
            self.conv_offset_aff.bias.data.zero_()
            if self.affinity == 'TC':
                self.aff_scale_const = nn.Parameter(self.num * torch.ones(1))
                self.aff_scale_const.requires_grad = False
            elif self.affinity == 'TGASS':
                self.aff_scale_const = nn.Parameter(self.args.affinity_gamma * self.num * torch.ones(1))
            else:
                self.aff_scale_const = nn.Parameter(torch.ones(1))
                self.aff_scale_const.requires_grad = False
        else:
            raise NotImplementedError
        # "dummy parameters for gathering" (:73-80)
        self.w = nn.Parameter(torch.ones((self.ch_f, 1, self.k_f, self.k_f)))
        self.b = nn.Parameter(torch.zeros(self.ch_f))
        self.w.requires_grad = False
        self.b.requires_grad = False
        self.w_conf = nn.Parameter(torch.ones((1, 1, 1, 1)))
        self.w_conf.requires_grad = False
        self.stride = 1
        self.padding = pad_f
        self.dilation = 1
        self.groups = self.ch_f
        self.deformable_groups = 1
        self.im2col_step = 64
        self.fuse_guidance_conv = True      # inference path: dd_nlspn_guided_offset_affinity where its geometry applies (A/B switch)

    # -- the reference's formulation (autograd path) ---------------------------------------------------------------------
    def _get_offset_affinity(self, guidance, confidence=None, rgb=None):
        B, _, H, W = guidance.shape
        offset_aff = self.conv_offset_aff(guidance)
        o1, o2, aff = torch.chunk(offset_aff, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1).view(B, self.num, 2, H, W)
        list_offset = list(torch.chunk(offset, self.num, dim=1))
        list_offset.insert(self.idx_ref, torch.zeros((B, 1, 2, H, W)).type_as(offset))
        offset = torch.cat(list_offset, dim=1).view(B, -1, H, W)
        if self.affinity in ['AS', 'ASS']:
            pass
        elif self.affinity == 'TC':
            aff = torch.tanh(aff) / self.aff_scale_const
        elif self.affinity == 'TGASS':
            aff = torch.tanh(aff) / (self.aff_scale_const + 1e-8)
        if self.args.conf_prop:
            list_conf = []
            offset_each = torch.chunk(offset, self.num + 1, dim=1)
            modulation_dummy = torch.ones((B, 1, H, W)).type_as(offset).detach()
            for idx_off in range(0, self.num + 1):
                ww = idx_off % self.k_f
                hh = idx_off // self.k_f
                if ww == (self.k_f - 1) / 2 and hh == (self.k_f - 1) / 2:
                    continue
                offset_tmp = offset_each[idx_off].detach()
                if self.args.legacy:            # in place on a view of `offset`, as in the reference (:126-134)
                    offset_tmp[:, 0, :, :] = offset_tmp[:, 0, :, :] + hh - (self.k_f - 1) / 2
                    offset_tmp[:, 1, :, :] = offset_tmp[:, 1, :, :] + ww - (self.k_f - 1) / 2
                conf_tmp = ModulatedDeformConvFunction.apply(confidence, offset_tmp.contiguous(), modulation_dummy, self.w_conf, self.b,
                                                             self.stride, 0, self.dilation, self.groups, self.deformable_groups,
                                                             self.im2col_step)
                list_conf.append(conf_tmp)
            conf_aff = torch.cat(list_conf, dim=1)
            aff = aff * conf_aff.contiguous()
        aff_abs = torch.abs(aff)
        aff_abs_sum = torch.sum(aff_abs, dim=1, keepdim=True) + 1e-4
        if self.affinity in ['ASS', 'TGASS']:
            aff_abs_sum[aff_abs_sum < 1.0] = 1.0
        if self.affinity in ['AS', 'ASS', 'TGASS']:
            aff = aff / aff_abs_sum
        aff_sum = torch.sum(aff, dim=1, keepdim=True)
        aff_ref = 1.0 - aff_sum
        list_aff = list(torch.chunk(aff, self.num, dim=1))
        list_aff.insert(self.idx_ref, aff_ref)
        aff = torch.cat(list_aff, dim=1)
        return offset, aff

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
        if not feat_init.is_cuda:
            raise RuntimeError("NLSPN runs only on a HIP device (the reference's DCN extension has no CPU path either: "
                               "src/model/deformconv/src/modulated_deform_conv.h:39-43)")
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
        # training: the reference's loop (:176-207) on the HIP DCNv2 operator
        offset, aff = self._get_offset_affinity(guidance, confidence if self.args.conf_prop else None, rgb)
        if self.args.preserve_input:
            mask_fix = torch.sum(feat_fix > 0.0, dim=1, keepdim=True).detach()
            mask_fix = (mask_fix > 0.0).type_as(feat_fix)
        feat_result = feat_init
        list_feat = []
        for k in range(1, self.prop_time + 1):
            if self.args.preserve_input:
                feat_result = (1.0 - mask_fix) * feat_result + mask_fix * feat_fix
            feat_result = self._propagate_once(feat_result, offset, aff)
            list_feat.append(feat_result)
        return feat_result, list_feat, offset, aff, self.aff_scale_const.data
