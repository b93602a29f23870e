"""Deterministic synthetic weights and inputs for the DDIM hot path.

There is no network on the build or GPU boxes, so neither the reference's datasets
nor its HuggingFace checkpoints are available.  Everything the tests, the golden
generator (tests/golden/make_golden.py), smoke() and bench.py feed to the kernels
comes from here.  The streams are drawn from ``numpy.random.RandomState`` (a frozen
legacy generator: the same seed gives the same bits on every numpy version/host), so a
fixture minted in the build container can be re-created bit-for-bit on the GPU box.

State-dict key names and shapes follow the reference head (prefix ``depth_head.`` there):
  * ``model.*``            ScheduledCNNRefine  (reference src/model/head/ddim_depth_estimate_res.py:300-322)
  * ``model.upsample_fuse.*``  Swin variant only (…/ddim_depth_estimate_res_swin_addHAHI.py:321-333)
  * ``depth_transform.*``  DeepDepthTransformWithUpsampling (reference src/model/ops/depth_transform.py:10-26)
"""
from __future__ import annotations

import numpy as np

LATENT_C = 16      # depth_feature_dim            (reference src/model/diffusion_dcbase_model.py:84)
COND_C = 256       # fpn_dim / channels_in        (reference …/ddim_depth_estimate_res.py:27-28)
HID_C = 64         # hidden width of the denoiser (reference …/ddim_depth_estimate_res.py:303,315)
GN_GROUPS = 4      # nn.GroupNorm(4, C)           (reference …/ddim_depth_estimate_res.py:305)
EMB_ROWS = 1280    # nn.Embedding(1280, 256)      (reference …/ddim_depth_estimate_res.py:313)


def _uniform(rs, shape, bound):
    return rs.uniform(-bound, bound, size=shape).astype(np.float32)


def _conv(rs, out_c, in_c, k, bias=True, gain=1.0):
    """PyTorch default Conv2d init scale: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weight and bias."""
    bound = gain / np.sqrt(in_c * k * k)
    w = _uniform(rs, (out_c, in_c, k, k), bound)
    b = _uniform(rs, (out_c,), bound) if bias else None
    return w, b


def _norm_affine(rs, c):
    # non-trivial affine so that a swapped gamma/beta or a dropped channel index shows up
    return (rs.uniform(0.6, 1.4, size=(c,)).astype(np.float32),
            rs.uniform(-0.3, 0.3, size=(c,)).astype(np.float32))


def _bn(rs, c):
    g, b = _norm_affine(rs, c)
    mean = rs.uniform(-0.2, 0.2, size=(c,)).astype(np.float32)
    var = rs.uniform(0.5, 1.5, size=(c,)).astype(np.float32)
    return g, b, mean, var


def make_state_dict(seed: int = 7240, variant: str = "res", decoder_gain: float = 0.05,
                    decoder_log_scale: float = 0.0) -> dict:
    """Return {key: float32 ndarray} for the hot-path submodules of one head.

    variant: "res" (ScheduledCNNRefine with skip-add) or "swin" (adds upsample_fuse.convA/convB).
    The decoder ends in 1/sigmoid(z) - 1 == exp(-z) (reference depth_transform.py:33-35); with an
    untrained denoiser the final latent has |x_0| ~ 1e2, so ``decoder_gain`` (scale of the last
    decoder conv, default 0.05) keeps z in a few units -> depths ~0.2..7 m, and
    ``decoder_log_scale`` = s multiplies every decoded depth by e^s (s=3 -> ~4..140 m, the
    KITTI far range where an absolute depth tolerance is hardest to meet).
    Default seed 7240 is the reference's default --seed (reference src/config.py:42-45).
    """
    rs = np.random.RandomState(seed)
    sd = {}

    def put_conv(prefix, out_c, in_c, k=3, bias=True, gain=1.0):
        w, b = _conv(rs, out_c, in_c, k, bias, gain)
        sd[prefix + ".weight"] = w
        if bias:
            sd[prefix + ".bias"] = b

    def put_gn(prefix, c):
        g, b = _norm_affine(rs, c)
        sd[prefix + ".weight"], sd[prefix + ".bias"] = g, b

    def put_bn(prefix, c):
        g, b, m, v = _bn(rs, c)
        sd[prefix + ".weight"], sd[prefix + ".bias"] = g, b
        sd[prefix + ".running_mean"], sd[prefix + ".running_var"] = m, v

    # --- denoiser --------------------------------------------------------------------
    put_conv("model.noise_embedding.0", HID_C, LATENT_C)
    put_gn("model.noise_embedding.1", HID_C)
    put_conv("model.noise_embedding.3", COND_C, HID_C)
    put_gn("model.noise_embedding.4", COND_C)
    sd["model.time_embedding.weight"] = rs.standard_normal((EMB_ROWS, COND_C)).astype(np.float32)
    put_conv("model.pred.0", HID_C, COND_C)
    put_gn("model.pred.1", HID_C)
    put_conv("model.pred.3", LATENT_C, HID_C)
    put_gn("model.pred.4", LATENT_C)
    if variant == "swin":
        put_conv("model.upsample_fuse.convA.conv", COND_C, COND_C)
        put_conv("model.upsample_fuse.convB.conv", COND_C, COND_C)
    elif variant != "res":
        raise ValueError(f"unknown variant {variant!r}")

    # --- latent encoder / decoder -------------------------------------------------------
    put_conv("depth_transform.conv_transform.0.0", LATENT_C, 1, bias=False)
    put_bn("depth_transform.conv_transform.0.1", LATENT_C)
    put_conv("depth_transform.conv_transform.1.0", LATENT_C, LATENT_C, bias=False)
    put_bn("depth_transform.conv_transform.1.1", LATENT_C)
    # ConvTranspose2d weight is (in, out, kH, kW); torch's fan_in for it is out*k*k
    bound = 1.0 / np.sqrt(LATENT_C * 16)
    sd["depth_transform.conv_inv_transform.0.weight"] = _uniform(rs, (LATENT_C, LATENT_C, 4, 4), bound)
    sd["depth_transform.conv_inv_transform.0.bias"] = _uniform(rs, (LATENT_C,), bound)
    put_bn("depth_transform.conv_inv_transform.1", LATENT_C)
    put_conv("depth_transform.conv_inv_transform.3.0", 1, LATENT_C, gain=decoder_gain)
    sd["depth_transform.conv_inv_transform.3.0.bias"] -= np.float32(decoder_log_scale)
    return sd


def latent_hw(H: int, W: int) -> tuple:
    """Latent size produced by the stride-2 k3 p1 encoder conv (reference depth_transform.py:16)."""
    return (H - 1) // 2 + 1, (W - 1) // 2 + 1


def make_inputs(seed: int, B: int, h: int, w: int, cond_hw=None) -> dict:
    """x_T ~ N(0,1) (reference …res.py:277), condition map |N(0,1)| (post-ReLU FPN output, >= 0),
    DDIM-loss noise ~ N(0,1) and per-sample timesteps (reference …res.py:203-207)."""
    rs = np.random.RandomState(seed)
    ch, cw = cond_hw if cond_hw is not None else (h, w)
    return {
        "x_T": rs.standard_normal((B, LATENT_C, h, w)).astype(np.float32),
        "cond": np.abs(rs.standard_normal((B, COND_C, ch, cw))).astype(np.float32),
        "noise": rs.standard_normal((B, LATENT_C, h, w)).astype(np.float32),
        "timesteps": rs.randint(0, 1000, size=(B,)).astype(np.int64),
    }


def make_gt_depth(seed: int, B: int, H: int, W: int, max_depth: float = 80.0, sparsity: float = 0.7) -> np.ndarray:
    """KITTI-like sparse ground-truth depth: U(0,max_depth) with ~70 % zeros (only its shape matters
    to the DDIM heads; reference …res.py:102,128)."""
    rs = np.random.RandomState(seed)
    d = rs.uniform(0.0, max_depth, size=(B, 1, H, W)).astype(np.float32)
    d[rs.uniform(size=d.shape) < sparsity] = 0.0
    return d


def make_fpn_state_dict(seed: int = 7241, in_channels=(64, 128, 256, 512), fpn_dim: int = COND_C) -> dict:
    """Weights of the condition-aggregation FPN of the Res head (conv_lateral / conv_up / the unused
    convup_fp; reference …/ddim_depth_estimate_res.py:42-84).  Stays in PyTorch; only needed to run
    the whole head end to end."""
    rs = np.random.RandomState(seed)
    sd = {}

    def put_bn(prefix, c):
        g, b, m, v = _bn(rs, c)
        sd[prefix + ".weight"], sd[prefix + ".bias"] = g, b
        sd[prefix + ".running_mean"], sd[prefix + ".running_var"] = m, v

    for i, c in enumerate(in_channels):
        w, _ = _conv(rs, fpn_dim, c, 3, bias=False)
        sd[f"conv_lateral.{i}.0.weight"] = w
        put_bn(f"conv_lateral.{i}.1", fpn_dim)
        if i != 0:
            bound = 1.0 / np.sqrt(fpn_dim * 4)
            sd[f"conv_up.{i - 1}.0.weight"] = _uniform(rs, (fpn_dim, fpn_dim, 2, 2), bound)
            put_bn(f"conv_up.{i - 1}.1", fpn_dim)
    sd["convup_fp.0.weight"] = _uniform(rs, (fpn_dim, fpn_dim, 2, 2), 1.0 / np.sqrt(fpn_dim * 4))
    put_bn("convup_fp.1", fpn_dim)
    return sd


def make_backbone_features(seed: int, B: int, H: int, W: int, in_channels=(64, 128, 256, 512)) -> list:
    """Stand-in for the backbone output: 4 non-negative maps at strides 2/4/8/16
    (reference src/model/backbone/mmbev_resnet.py:101-160 emits [64,128,256,512] channels)."""
    rs = np.random.RandomState(seed)
    feats = []
    h, w = H, W
    for c in in_channels:
        h, w = (h + 1) // 2, (w + 1) // 2
        feats.append(np.abs(rs.standard_normal((B, c, h, w))).astype(np.float32))
    return feats


def make_hahi_state_dict(seed: int = 7242, in_channels=(192, 384, 768, 1536), embedding_dim: int = 512, num_points: int = 8,
                         prefix: str = "hahineck.") -> dict:
    """Weights of HAHIHeteroNeck as the heads build it (reference src/model/necks/hahi.py:56-118; in_channels == out_channels,
    src/model/head/ddim_depth_estimate_res_swin_addHAHI.py:54-56): ConvModule = conv (no bias) + bn, plus the never-executed
    attention / reference-point / level-embedding parameters."""
    rs = np.random.RandomState(seed)
    sd = {}

    def conv_module(name, cin, cout, k):
        w, _ = _conv(rs, cout, cin, k, bias=False, gain=1.7)       # a little above the default scale: ReLU halves the variance
        sd[prefix + name + ".conv.weight"] = w
        g, b, m, v = _bn(rs, cout)
        sd[prefix + name + ".bn.weight"], sd[prefix + name + ".bn.bias"] = g, b
        sd[prefix + name + ".bn.running_mean"], sd[prefix + name + ".bn.running_var"] = m, v

    def linear(name, cin, cout):
        sd[prefix + name + ".weight"] = _uniform(rs, (cout, cin), 1.0 / np.sqrt(cin))
        sd[prefix + name + ".bias"] = _uniform(rs, (cout,), 1.0 / np.sqrt(cin))

    for i, c in enumerate(in_channels):
        conv_module(f"lateral_convs.{i}", c, c, 1)
    for i, c in enumerate(in_channels[1:]):
        conv_module(f"trans_proj.{i}", c, embedding_dim, 1)
    for i, c in enumerate(in_channels[1:]):
        conv_module(f"trans_fusion.{i}", c + embedding_dim, c, 3)
    conv_module("conv_proj.0", in_channels[0], embedding_dim, 1)
    conv_module("conv_fusion.0", in_channels[0] + embedding_dim, in_channels[0], 3)
    linear("reference_points", embedding_dim, 2)
    sd[prefix + "level_embed"] = rs.standard_normal((4, embedding_dim)).astype(np.float32)
    for att in ("multi_att", "self_attn"):
        linear(att + ".sampling_offsets", embedding_dim, 8 * 4 * num_points * 2)
        linear(att + ".attention_weights", embedding_dim, 8 * 4 * num_points)
        linear(att + ".value_proj", embedding_dim, embedding_dim)
        linear(att + ".output_proj", embedding_dim, embedding_dim)
    return sd
