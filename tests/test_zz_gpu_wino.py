"""GPU, OPT-IN (`DD_TEST_WINOGRAD=1 pytest -m gpu tests/test_zz_gpu_wino.py`): the experimental Winograd F(2x2,3x3) kernel for the Swin
denoiser's convB (csrc/dd_wino.hip, option "winograd").  Written at the end of round 1 with no GPU time left: its index math is pinned on
the CPU (tests/test_wino_kernel_emulation.py), its numerics by tools/winograd_numerics.py; this file is what the first GPU run of round 2
should execute.  Skipped by default so that an unvalidated kernel cannot fail the round-end suite.
State at the end of round 1 (tools/gpu/wino_try.py, run 40): v1 ran once -- eps error 1.09x the direct kernel's in f16 and bf16, 1131 us per
KITTI B=4 launch against 494 us direct; v2 (double-buffered, MFMAs overlapping the next chunk's transform) has never run."""
import os

import numpy as np
import pytest
import torch

from diffusiondepth_amd import synth

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("DD_TEST_WINOGRAD") != "1", reason="experimental kernel: set DD_TEST_WINOGRAD=1")]


@pytest.mark.parametrize("version", [1, 2, 3], ids=["v1", "v2_double_buffered", "v3_packed_f16_transform"])
@pytest.mark.parametrize("prec", ["f16", "bf16"])
@pytest.mark.parametrize("shape", [(1, 16, 24), (2, 13, 21), (1, 44, 152)], ids=["small", "ragged_b2", "kitti_quarter"])
def test_winograd_convB_matches_the_direct_kernel(prec, shape, version):
    import diffusiondepth_amd as dda
    import gpu_util as U
    B, h, w = shape
    be = dda.HipDenoiser(variant="swin")
    be.load_state_dict(synth.make_state_dict(7240, "swin"))
    be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
    inp = synth.make_inputs(5, B, h, w, ((h + 1) // 2, (w + 1) // 2))
    x, cond = U.cu(inp["x_T"]), U.cu(inp["cond"])
    t = U.cu(inp["timesteps"])
    ref = be.denoise_once(x, t, cond, "fp32").cpu().numpy()
    be.set_option("winograd", 0)
    direct = be.denoise_once(x, t, cond, prec).cpu().numpy()
    be.set_option("winograd", version)
    wino = be.denoise_once(x, t, cond, prec).cpu().numpy()
    e_d, e_w = U.rms(direct, ref), U.rms(wino, ref)
    U.record(f"winograd_convB_{prec}_{h}x{w}", eps_rms_direct=e_d, eps_rms_winograd=e_w, eps_max=float(np.abs(ref).max()))
    assert np.isfinite(wino).all()
    assert e_w < 2.0 * e_d + 1e-4, (e_d, e_w)             # tools/winograd_numerics.py: ~1.2x the direct kernels' error
    # the whole loop, graph replay included
    T = 5
    x0_d = be.denoise(x, cond, T, prec)
    be.set_option("winograd", 0)
    x0_ref = be.denoise(x, cond, T, "fp32")
    x0_dir = be.denoise(x, cond, T, prec)
    s = float(x0_ref.abs().max())
    assert float((x0_d - x0_ref).abs().max()) < 2.0 * float((x0_dir - x0_ref).abs().max()) + 1e-4 * s


@pytest.mark.parametrize("dma", [0, 1], ids=["regs", "lds_dma"])
@pytest.mark.parametrize("variant", ["res", "swin"])
@pytest.mark.parametrize("prec,opt", [("f16", 4), ("f16", 5), ("bf16", 4)], ids=["f16", "f16_packed_transform", "bf16"])
def test_winograd_all_large_convolutions(variant, prec, opt, dma):
    """Option 4 / 5: conv2 + conv3 (Res) or conv2 + convA + convB + pred.0 (Swin) on the generalised double-buffered kernel, with the
    GroupNorm (+ condition) prologue from wino_gn_table_kernel and the GroupNorm partial sums in the epilogue.  NEVER RUN in round 1."""
    import diffusiondepth_amd as dda
    import gpu_util as U
    B, h, w = 2, 21, 45
    be = dda.HipDenoiser(variant=variant)
    be.load_state_dict(synth.make_state_dict(7240, variant))
    be.set_schedule(dda.DDIMScheduler().alphas_cumprod)
    inp = synth.make_inputs(9, B, h, w, ((h + 1) // 2, (w + 1) // 2) if variant == "swin" else None)
    x, cond, t = U.cu(inp["x_T"]), U.cu(inp["cond"]), U.cu(inp["timesteps"])
    ref = be.denoise_once(x, t, cond, "fp32").cpu().numpy()
    direct = be.denoise_once(x, t, cond, prec).cpu().numpy()
    be.set_option("winograd_dma", dma)
    be.set_option("winograd", opt)
    wino = be.denoise_once(x, t, cond, prec).cpu().numpy()
    e_d, e_w = U.rms(direct, ref), U.rms(wino, ref)
    U.record(f"winograd_all_{variant}_{prec}_opt{opt}_dma{dma}", eps_rms_direct=e_d, eps_rms_winograd=e_w)
    assert np.isfinite(wino).all() and e_w < 2.5 * e_d + 1e-4, (e_d, e_w)
    T = 5
    x0_w = be.denoise(x, cond, T, prec)
    be.set_option("winograd", 0)
    x0_ref, x0_d = be.denoise(x, cond, T, "fp32"), be.denoise(x, cond, T, prec)
    s = float(x0_ref.abs().max())
    assert float((x0_w - x0_ref).abs().max()) < 2.5 * float((x0_d - x0_ref).abs().max()) + 1e-4 * s
