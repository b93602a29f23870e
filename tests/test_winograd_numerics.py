"""CPU: pins the numerics claim behind DESIGN.md section 7 item 0 (Winograd F(2x2,3x3) for conv2 / conv3 as the next kernel): the
emulation of the 16-bit kernels' rounding points in tools/winograd_numerics.py, on one small case.
  * the Winograd formulation in fp32 is the same convolution (1e-5 relative on a single layer, decoded depth ~1e-6 over the loop);
  * with f16 operands its depth error stays inside the 1e-3 class and within 1.6x of the direct kernels' emulated error."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import winograd_numerics as WN  # noqa: E402
from diffusiondepth_amd import synth  # noqa: E402
from oracle import torch_cpu_port as P  # noqa: E402


def test_winograd_convolution_equals_direct_in_fp32():
    rs = np.random.RandomState(0)
    for (C, Co, H, W) in ((8, 5, 7, 10), (16, 16, 12, 9)):           # odd and even sizes
        a = torch.from_numpy(rs.standard_normal((2, C, H, W)).astype(np.float32))
        w = torch.from_numpy(rs.standard_normal((Co, C, 3, 3)).astype(np.float32))
        b = torch.from_numpy(rs.standard_normal(Co).astype(np.float32))
        d, y = WN.conv_direct(a, w, b, None), WN.conv_winograd(a, w, b, None)
        assert float((d - y).abs().max() / d.abs().max()) < 1e-5


def test_winograd_f16_loop_error_stays_in_the_1e3_class():
    torch.set_num_threads(4)
    sd = P.to_torch_sd(synth.make_state_dict(7240))
    inp = synth.make_inputs(100, 1, 16, 24)
    T = 10
    ref = P.decode(sd, P.ddim_loop(sd, inp["x_T"], inp["cond"], T))
    rmse = lambda d: float((d - ref).pow(2).mean().sqrt())
    e_fp32 = rmse(P.decode(sd, WN.loop(sd, inp["x_T"], inp["cond"], T, None, True)))
    e_dir = rmse(P.decode(sd, WN.loop(sd, inp["x_T"], inp["cond"], T, torch.float16, False)))
    e_win = rmse(P.decode(sd, WN.loop(sd, inp["x_T"], inp["cond"], T, torch.float16, True)))
    assert e_fp32 < 5e-6
    assert e_win < 1e-3 and e_win < 1.6 * e_dir + 1e-5, (e_dir, e_win)
