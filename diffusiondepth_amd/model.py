"""Diffusion_DCbase_Model facade (reference src/model/diffusion_dcbase_model.py:26-224): same
``forward(sample) -> dict`` contract, same ``args`` attributes (backbone_module/backbone_name/
head_specify/inference_steps/num_train_timesteps).  The visual backbone stays in PyTorch-ROCm; a
stem-less BasicBlock ResNet equivalent of the reference's mmbev_res18/50/101
(reference src/model/backbone/mmbev_resnet.py:101-194, which needs mmdet) is provided so the whole
model can run offline with random-init weights.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch
from torch import nn

from .head import build_head


class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride, first=False):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        # the reference passes a bare 3x3 conv (with bias) as the first block's downsample (mmbev_resnet.py:128-129)
        self.downsample = nn.Conv2d(cin, cout, 3, stride, 1) if first else None

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class ResNetForMMBEV(nn.Module):
    """Stem-less BasicBlock stacks emitting [64,128,256,512] channels at strides 2/4/8/16
    (reference mmbev_resnet.py:101-160)."""

    def __init__(self, num_layer, in_channels=3, channels=(64, 128, 256, 512), strides=(2, 2, 2, 2)):
        super().__init__()
        layers = []
        cin = in_channels
        for n, c, s in zip(num_layer, channels, strides):
            blocks = [_BasicBlock(cin, c, s, first=True)] + [_BasicBlock(c, c, 1) for _ in range(n - 1)]
            layers.append(nn.Sequential(*blocks))
            cin = c
        self.layers = nn.Sequential(*layers)

    def forward(self, x):
        feats = []
        for layer in self.layers:
            x = layer(x)
            feats.append(x)
        return feats


BACKBONES = {                                       # reference mmbev_resnet.py:176-194
    "mmbev_res18": lambda: ResNetForMMBEV([2, 2, 2, 2]),
    "mmbev_res50": lambda: ResNetForMMBEV([3, 4, 6, 3]),      # BasicBlock [3,4,6,3]: the ResNet-34 layout
    "mmbev_res101": lambda: ResNetForMMBEV([3, 4, 23, 3]),
}


def default_args(**over):
    """The subset of reference src/config.py flags the model reads."""
    a = dict(backbone_module="mmbev_resnet", backbone_name="mmbev_res50", head_specify="DDIMDepthEstimate_Res",
             inference_steps=20, num_train_timesteps=1000, model_name="Diffusion_DCbase_", precision=None)
    a.update(over)
    return SimpleNamespace(**a)


def resolve_backbone(args):
    """The reference's rule (src/model/backbone/__init__.py:5-11): ``getattr(import_module('model.backbone.' +
    args.backbone_module.lower()), args.backbone_name)`` -- a zero-argument factory.  The bundled mmbev_res* family is served from
    BACKBONES (no mmdet needed); any other name is imported from ``args.backbone_package`` (default 'model.backbone', i.e. the
    reference tree this package is dropped into: swin.py / depthformerswin.py / mpvit.py stay upstream PyTorch per the north star)."""
    from importlib import import_module
    name = args.backbone_name
    if name in BACKBONES:
        return BACKBONES[name]
    pkg = getattr(args, "backbone_package", None) or "model.backbone"
    modname = f"{pkg}.{str(args.backbone_module).lower()}"
    try:
        module = import_module(modname)
    except ImportError as e:
        raise ImportError(f"backbone {name!r}: cannot import {modname!r} ({e}); only the mmbev_res* family is bundled -- run inside the "
                          "reference tree (its src/ on sys.path), set args.backbone_package, or pass depth_backbone=<nn.Module>") from e
    return getattr(module, name)


class Diffusion_DCbase_Model(nn.Module):
    def __init__(self, args=None, depth_backbone=None, depth_head=None, **kwargs):
        """As the reference (diffusion_dcbase_model.py:28-91): ``depth_backbone`` / ``depth_head`` may be injected as built modules;
        otherwise the backbone is resolved from args.backbone_module / args.backbone_name and the head from args.head_specify.  The
        head classes fix their own pyramid widths (the reference passes in_channels=[64,128,256,512] to every head and each head
        overrides it, ...res.py:31, ...res_swin_add.py:31), so any backbone emitting that head's 4 maps composes."""
        super().__init__()
        self.args = args if args is not None else default_args()
        self.depth_backbone = depth_backbone if depth_backbone is not None else resolve_backbone(self.args)()
        if depth_head is not None:
            self.depth_head = depth_head
        else:
            self.depth_head = build_head(dict(type=self.args.head_specify, in_channels=[64, 128, 256, 512],
                                              inference_steps=getattr(self.args, "inference_steps", 20),
                                              num_train_timesteps=getattr(self.args, "num_train_timesteps", 1000), depth_feature_dim=16,
                                              loss_cfgs=[], init_cfg=None, precision=getattr(self.args, "precision", None)))

    def extract_depth(self, img, depth_map, depth_mask, gt_depth_map=None, return_loss=False, **kw):
        fp = self.depth_backbone(img)                                                   # :125
        return self.depth_head(fp, depth_map, depth_mask, gt_depth_map=gt_depth_map, return_loss=return_loss, image=img, **kw)

    def forward(self, sample):
        """sample keys rgb (B,3,H,W), gt (B,1,H,W), dep, depth_map, depth_mask (reference :186-224)."""
        rgb, gt = sample["rgb"], sample["gt"]
        depth_map = sample.get("depth_map", sample.get("dep"))
        depth_mask = sample.get("depth_mask")
        return self.extract_depth(rgb, depth_map, depth_mask, gt_depth_map=gt, return_loss=self.training,
                                  sparse_depth=sample.get("dep"))
