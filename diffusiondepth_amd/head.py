"""DDIMDepthEstimate_Res: drop-in mirror of the reference head
(reference src/model/head/ddim_depth_estimate_res.py:14-235).

Same constructor keywords (the dict Diffusion_DCbase_Model passes to HEADS.build,
reference src/model/diffusion_dcbase_model.py:77-91), same parameter tree (state_dict keys), same
forward(fp, depth_map, depth_mask, gt_depth_map=None, return_loss=False, **kwargs) -> 13-key dict.
The latent encoder, the condition aggregation (conv_lateral / conv_up FPN, …res.py:108-118; Res head, eval
mode), the T-step DDIM loop, the decoder and the ddim_loss denoiser call run in the HIP library.  mmcv/mmdet3d are not needed: the few factories the reference pulls from them are plain
torch.nn layers.
"""
from __future__ import annotations

import os

import torch
import torch.nn.functional as F
from torch import nn

from . import modules as _modules
from .modules import CNNDDIMPipiline, CNNDDIMPipilineVis, DeepDepthTransformWithUpsampling, HipBound, ScheduledCNNRefine
from .necks import HAHIHeteroNeck
from .scheduler import DDIMScheduler

HEADS = {}          # name -> class; stands in for the mmdet3d HEADS registry (…res.py:14)

# The two shipped configurations of a head (constructor keyword `profile`, environment DDEPTH_PROFILE).  "reference" is the DEFAULT: a drop-in
# reproduces the reference's numbers and RNG streams; "fast" is the configuration bench.py's headline and `head_forward` extra time.
PROFILES = {"reference": {"precision": "fp32", "loss_noise_device": "cpu"},
            "fast": {"precision": "f16r", "loss_noise_device": "device"}}


def register_head(cls):
    HEADS[cls.__name__] = cls
    return cls


def build_head(cfg: dict):
    cfg = dict(cfg)
    return HEADS[cfg.pop("type")](**cfg)


@register_head
class DDIMDepthEstimate_Res(nn.Module):
    _IN_CHANNELS = [64, 128, 256, 512]       # the reference overrides the constructor argument with these (…res.py:31)
    _VARIANT = "res"
    _HAHI = False                            # HAHIHeteroNeck in front of the FPN (…swin_addHAHI.py:54-56,110)
    _VIS = False                             # *Vis heads: pipeline returns every intermediate sample (…res_vis.py:124,141-143)
    _HIP_FPN_WIDTHS = True                   # dd_condition is specialised for the ResNet / Swin-L pyramid widths

    def __init__(self, in_channels=(64, 128, 256, 512), up_scale_factor=1, inference_steps=20, num_train_timesteps=1000,
                 return_indices=None, depth_transform_cfg=None, depth_feature_dim=16, detach_fp=False, loss_cfgs=(),
                 init_cfg=None, precision=None, condition_backend="hip", eval_ddim_loss=True, loss_noise_device=None, profile=None, **kwargs):
        """Beyond the reference's keywords (src/model/diffusion_dcbase_model.py:77-91):
        profile            the two shipped configurations of a head (PROFILES below; None = $DDEPTH_PROFILE, else "reference"):
                             "reference" [default]  what the reference does, bit for bit where that is defined: fp32 arithmetic (the reference runs
                                                    apex O0), the loss noise drawn on the HOST generator (…res.py:203) -- the same RNG streams, so a
                                                    seeded evaluation reproduces the reference's x_T, timesteps and `ddim_loss` forward after forward
                             "fast"                 the configuration bench.py headlines: precision "f16r" (depth RMSE vs the reference <= 1e-3 / 1.5 at
                                                    KITTI's range; Swin: the single ddim_loss call in f16, see ScheduledCNNRefine.forward) and the loss
                                                    noise drawn on the device from a PRIVATE generator (the global device stream x_T / timesteps come
                                                    from is not advanced, so `pred` of every forward still reproduces the reference's draws)
                           an explicit `precision=` / `loss_noise_device=` overrides the profile's choice of that one setting.
        precision          operand mode of the HIP kernels ("fp32" / "f16x3" abs-grade parity, "f16r" / "f16" / "bf16"; None = the profile's)
        condition_backend  "hip" (dd_condition) or "torch" for the FPN
        eval_ddim_loss     True = the reference's behaviour: the DDIM loss (one more denoiser call, fed by a CPU randn of the latent's
                           size) is computed on EVERY forward, also in eval / test (…res.py:159-169).  False skips it
                           outside .train() and returns a zero scalar under 'ddim_loss' (inference-only deployments).
        loss_noise_device  "cpu" = draw the loss noise on the host and copy it over, as the reference does (…res.py:203, quirk q3: same
                           RNG stream as the reference); "device" = draw it on the GPU from a private generator of this head (same
                           distribution; saves the 6.8 MB-per-KITTI-map host RNG + H2D copy per forward: 15.6 -> 8.0 ms per KITTI eval
                           forward); "auto" = "cpu" in .train() -- the reference's training RNG stream -- and "device" in eval."""
        super().__init__()
        profile = profile or os.environ.get("DDEPTH_PROFILE") or "reference"      # (an empty DDEPTH_PROFILE means "not set")
        if profile not in PROFILES:
            raise ValueError(f"profile must be one of {sorted(PROFILES)}")
        self.profile = profile
        if precision is None:
            precision = os.environ.get("DDEPTH_PRECISION") or PROFILES[profile]["precision"]
        if loss_noise_device is None:
            loss_noise_device = PROFILES[profile]["loss_noise_device"]
        if loss_noise_device not in ("cpu", "device", "auto"):
            raise ValueError("loss_noise_device must be 'cpu', 'device' or 'auto'")
        self._loss_noise_gen = None       # private device generator of the "device" route (created on first use)
        self._loss_noise_seed = None      # torch.initial_seed() it was seeded from
        # HAHI heads only: True runs the PyTorch neck under autocast in the kernels' 16-bit type.  Default False = the reference's fp32
        # arithmetic (the reference never autocasts the neck)
        self.neck_autocast = bool(kwargs.pop("neck_autocast", False))
        self.eval_ddim_loss = bool(eval_ddim_loss)
        self.loss_noise_device = loss_noise_device
        if depth_transform_cfg is not None and depth_transform_cfg.get("type", "DeepDepthTransformWithUpsampling") != \
                "DeepDepthTransformWithUpsampling":
            raise NotImplementedError("only DeepDepthTransformWithUpsampling is used by the reference heads (…res.py:23)")
        fpn_dim = 256
        in_channels = list(self._IN_CHANNELS)
        self.detach_fp = detach_fp
        self.loss_cfgs = list(loss_cfgs)
        self.init_cfg = init_cfg
        self.return_indices = return_indices
        self.up_scale = nn.Identity() if up_scale_factor == 1 else \
            (lambda t: F.interpolate(t, scale_factor=up_scale_factor, mode="bilinear"))
        bound = HipBound(self._VARIANT)
        self._bound = bound
        self.depth_transform = DeepDepthTransformWithUpsampling(hidden=16, eps=1e-6, bound=bound)
        self.model = ScheduledCNNRefine(channels_in=fpn_dim, channels_noise=depth_feature_dim, bound=bound, precision=precision,
                                        variant=self._VARIANT)
        self.diffusion_inference_steps = inference_steps
        self.scheduler = DDIMScheduler(num_train_timesteps=num_train_timesteps, clip_sample=False)
        self.pipeline = (CNNDDIMPipilineVis if self._VIS else CNNDDIMPipiline)(self.model, self.scheduler)
        if self._HAHI:
            self.hahineck = HAHIHeteroNeck(in_channels=list(in_channels), out_channels=list(in_channels), embedding_dim=512,
                                           positional_encoding=dict(type="SinePositionalEncoding", num_feats=256),
                                           scales=[1, 1, 1, 1], cross_att=False, self_att=False, num_points=8)
        else:
            # present in the reference state_dict although unused in forward (…res.py:42-52); the HAHI heads do not have it
            self.convup_fp = nn.Sequential(nn.ConvTranspose2d(fpn_dim, fpn_dim, 2, 2, bias=False), nn.BatchNorm2d(fpn_dim), nn.ReLU(True))
        self.conv_lateral = nn.ModuleList()
        self.conv_up = nn.ModuleList()
        for i, c in enumerate(in_channels):
            self.conv_lateral.append(nn.Sequential(nn.Conv2d(c, fpn_dim, 3, 1, 1, bias=False), nn.BatchNorm2d(fpn_dim), nn.ReLU(True)))
            if i != 0:
                self.conv_up.append(nn.Sequential(nn.ConvTranspose2d(fpn_dim, fpn_dim, 2, 2, bias=False), nn.BatchNorm2d(fpn_dim),
                                                  nn.ReLU(True)))
        if condition_backend not in ("hip", "torch"):
            raise ValueError("condition_backend must be 'hip' or 'torch'")
        self._hip_fpn = condition_backend == "hip" and self._HIP_FPN_WIDTHS
        if self._hip_fpn:
            bound.register("conv_lateral.", self.conv_lateral)
            bound.register("conv_up.", self.conv_up)
        # the HAHI neck's convolutions run in the library too (dd_neck_condition) for the two pyramids its kernels are built for (Swin-L,
        # MPViT-small); any other widths keep the PyTorch neck in front of the library's FPN
        self._hip_neck = (self._hip_fpn and self._HAHI and list(in_channels) in ([192, 384, 768, 1536], [128, 216, 288, 288])
                          and not (self.hahineck.cross_att or self.hahineck.self_att))        # (dd_neck_condition is the attention-off neck: what every head builds)
        if self._hip_neck:
            bound.register("hahineck.", self.hahineck)

    @staticmethod
    def _on_hip(tensors) -> bool:
        return all(t.is_cuda for t in tensors)

    # -- condition aggregation (…res.py:108-118) --------------------------------------------------
    def _library_fpn_ok(self, fp) -> bool:
        """dd_condition / dd_neck_condition are inference-only: not when autograd has to reach the backbone features or the FPN / neck
        weights (e.g. fine-tuning with the head in .eval() for frozen BatchNorm) -- never a silently dropped gradient."""
        mods = [self.conv_lateral, self.conv_up] + ([self.hahineck] if self._HAHI else [])
        needs_grad = torch.is_grad_enabled() and (any(f.requires_grad for f in fp) or any(p.requires_grad for m in mods for p in m.parameters()))
        return (self._hip_fpn and not self.training and not needs_grad and len(fp) == 4 and self._on_hip(fp)
                and self.model.precision != "naive_fp32")

    def aggregate_condition(self, fp, neck_in_library=False):
        if self._library_fpn_ok(fp):
            # eval-mode BatchNorm is folded into the convolutions inside the library (dd_condition / dd_neck_condition)
            be = self._bound.ensure(fp[0].device, self.scheduler, need=("fpn",))
            return be.condition([f.float() for f in fp], self.model.precision, neck=neck_in_library)
        assert not neck_in_library
        x = None
        n = len(fp)
        for i in range(n):
            f = fp[n - i - 1]
            cur = self.conv_lateral[n - i - 1](f)
            if i > 0:
                cur = cur + F.adaptive_avg_pool2d(self.conv_up[n - i - 1](x), output_size=cur.shape[-2:])
            x = cur
        return x

    def forward(self, fp, depth_map, depth_mask, gt_depth_map=None, return_loss=False, **kwargs):
        # parameters do not change inside one forward: verify / upload them once, not in front of each of its ~7 library calls
        with self._bound.hold():
            return self._forward(fp, depth_map, depth_mask, gt_depth_map=gt_depth_map, return_loss=return_loss, **kwargs)

    def _forward(self, fp, depth_map, depth_mask, gt_depth_map=None, return_loss=False, **kwargs):
        if self.detach_fp is not False and self.detach_fp is not None:
            if isinstance(self.detach_fp, (list, tuple, range)):
                fp = list(fp)
                for i in self.detach_fp:
                    fp[i] = fp[i].detach()
            else:
                fp = [it.detach() for it in fp]
        gt_map_t = self.depth_transform.t(gt_depth_map)                         # …res.py:102  (HIP encoder)
        neck_in_library = False
        if self._HAHI:
            if self._hip_neck and self._library_fpn_ok(fp):
                neck_in_library = True                                          # …swin_addHAHI.py:110 runs inside dd_neck_condition
            else:
                # PyTorch-ROCm neck (training, MPViT widths, autograd): fp32 as the reference, or under autocast when asked for
                ac = {"bf16": torch.bfloat16, "f16": torch.float16}.get(self.model.precision) if (self.neck_autocast and fp[0].is_cuda and not self.training) else None
                if ac is not None:
                    with torch.autocast("cuda", dtype=ac):
                        fp = [f.float() for f in self.hahineck(fp)]
                else:
                    fp = self.hahineck(fp)                                      # …swin_addHAHI.py:110
        x = self.aggregate_condition(fp, neck_in_library)                       # …res.py:108-118
        res = self.pipeline(batch_size=x.shape[0], device=x.device, dtype=x.dtype, shape=gt_map_t.shape[-3:],
                            input_args=(x, None, None, None),
                            num_inference_steps=self.diffusion_inference_steps, return_dict=False)                # :124-138
        refined_depth_t = res[0]
        refined_depth = self.depth_transform.inv_t(refined_depth_t)             # :140  (HIP decoder)
        # *Vis heads: every intermediate sample decoded (…res_vis.py:141-143)
        processes_vis = [self.depth_transform.inv_t(m) for m in res[1]] if self._VIS else None
        if self.training or self.eval_ddim_loss:
            ddim_loss = self.ddim_loss(pred_depth=refined_depth, gt_depth=gt_map_t, refine_module_inputs=(x, None, None, None),
                                       blur_depth_t=refined_depth_t, weight=1.0)    # :159-169
        else:
            ddim_loss = refined_depth.new_zeros(())
        return {"pred": refined_depth, "pred_init": gt_map_t, "blur_depth_t": gt_map_t, "ddim_loss": ddim_loss,
                "gt_map_t": gt_map_t, "pred_uncertainty": None, "pred_inter": processes_vis, "weight_map": None, "guidance": None,
                "offset": None, "aff": None, "gamma": None, "confidence": None}

    def ddim_loss(self, gt_depth, refine_module_inputs, blur_depth_t, weight, **kwargs):
        """…res.py:201-217: same RNG draw order (CPU randn for the noise, device randint for t)."""
        on_host = self.loss_noise_device == "cpu" or (self.loss_noise_device == "auto" and self.training) or not blur_depth_t.is_cuda
        if on_host:
            noise = torch.randn(blur_depth_t.shape).to(blur_depth_t.device)
        else:
            # a PRIVATE generator: the global device stream -- x_T (…res.py:277) and `timesteps` (:207) are drawn from it -- is not advanced, so the
            # reference's draws of this and every later forward stay what they are.  Seeded from torch's seed: reproducible runs.
            # (Re-)seeded whenever torch's seed changed since it was built -- a later torch.manual_seed() makes the loss reproducible again -- with the
            # process-group rank mixed in (data-parallel ranks that share a seed draw different noise, as ranks that call torch.randn on different
            # shards would).  The generator's state is not part of state_dict(): a resumed run restarts this stream.
            gen = self._loss_noise_gen
            seed_now = torch.initial_seed()
            if gen is None or gen.device != blur_depth_t.device or self._loss_noise_seed != seed_now:
                rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
                gen = self._loss_noise_gen = torch.Generator(device=blur_depth_t.device)
                gen.manual_seed((seed_now + 0x9E3779B97F4A7C15 + 0x632BE59BD9B4E019 * rank) & 0x7FFFFFFFFFFFFFFF)
                self._loss_noise_seed = seed_now
            noise = torch.randn(blur_depth_t.shape, device=blur_depth_t.device, generator=gen).to(blur_depth_t.device)     # (.to: a no-op unless a test injects its own draw)
        bs = blur_depth_t.shape[0]
        timesteps = torch.randint(0, self.scheduler.num_train_timesteps, (bs,), device=gt_depth.device).long()
        # the loop output is not detached in the reference: when it carries gradient, q_sample stays a torch op so that
        # autograd reaches the loop through it; otherwise dd_add_noise (HIP tensors; CPU tensors: the scheduler's torch mirror)
        keep_graph = (torch.is_grad_enabled() and blur_depth_t.requires_grad) or not _modules._on_hip_device(blur_depth_t)
        be = None if keep_graph else self._bound.ensure(blur_depth_t.device, self.scheduler, need=())        # q_sample needs the schedule only
        noisy_images = self.scheduler.add_noise(blur_depth_t, noise, timesteps, backend=None if keep_graph else be)
        noise_pred = self.model(noisy_images, timesteps, *refine_module_inputs)
        return F.mse_loss(noise_pred, noise)


@register_head
class DDIMDepthEstimate_Swin_ADD(DDIMDepthEstimate_Res):
    """Swin-L head (reference src/model/head/ddim_depth_estimate_res_swin_add.py:14-200): identical glue, backbone
    channels [192,384,768,1536], and the UpSample_add denoiser whose stride-4 condition map is bilinearly upsampled
    to the latent size inside the library (once per call: up(feat + E[t]) = up(feat) + E[t])."""
    _IN_CHANNELS = [192, 384, 768, 1536]
    _VARIANT = "swin"


@register_head
class DDIMDepthEstimate_Swin_ADDHAHI(DDIMDepthEstimate_Swin_ADD):
    """The head of the reference's headline Swin-L configuration (README.md:215,257; src/model/head/ddim_depth_estimate_res_swin_addHAHI.py:
    15-241): Swin_ADD with the HAHIHeteroNeck (attention off) in front of the condition FPN and without the unused ``convup_fp``."""
    _HAHI = True


@register_head
class DDIMDepthEstimate_MPVIT_ADDHAHI(DDIMDepthEstimate_Swin_ADDHAHI):
    """MPViT-small pyramid [128, 216, 288, 288] (src/model/head/ddim_depth_estimate_res_mpvit_HAHI.py:32,51).  Same UpSample_add
    denoiser; dd_condition carries the 216-channel level as 224 (zero channels / zero weights)."""
    _IN_CHANNELS = [128, 216, 288, 288]


@register_head
class DDIMDepthEstimate_ResVis(DDIMDepthEstimate_Res):
    """src/model/head/ddim_depth_estimate_res_vis.py: 'pred_inter' = every intermediate sample of the loop, decoded."""
    _VIS = True


@register_head
class DDIMDepthEstimate_Swin_ADDHAHIVis(DDIMDepthEstimate_Swin_ADDHAHI):
    """src/model/head/ddim_depth_estimate_res_swin_addHAHI_vis.py."""
    _VIS = True
