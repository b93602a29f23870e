"""The PRODUCTION convolution kernels (diffusiondepth_amd/csrc/dd_igemm2.hip: every layer of both denoisers, the FPN laterals, the data-gradient
layers) executed on the CPU from their own source (tests/host_emul), against an fp64 convolution.

These kernels are parity-green on the MI355X in hundreds of cases -- but a GPU run only shows that the hardware's timing did not trigger a
race.  Here the timing is adversarial and deterministic:
  * a wave runs ahead of the others as far as the workgroup barriers allow (first wave / last wave: `order`),
  * an LDS-DMA of a weight stage lands at issue (earliest: it must not overwrite a ring slot somebody still reads) or only when an
    s_waitcnt retires it (latest: `s_waitcnt vmcnt(NRAW)` must really cover it -- the counted-wait arithmetic of the kernel, with the raw-patch
    loads that stay in flight modelled in the wave's VMEM queue, dd_gcn.h),
and the results must be the same in all four combinations.  Test infrastructure only."""
from __future__ import annotations

import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hostemu_util import EK_BF16, EK_F16, GN_GROUPS, STAT_SLOTS, STAT_STRIDE, bind_igemm2, blocked, build_library, build_mutant, from16, ptr, to16, unblocked

EK_F32 = 0
P = ctypes.c_void_p


@pytest.fixture(scope="module")
def emu():
    return bind_igemm2(build_library())


def esz(ek):
    return 4 if ek == EK_F32 else 2


def enc(x, ek):
    """float32 array -> the element kind's storage (float32 itself or 16-bit patterns)"""
    return np.ascontiguousarray(x, np.float32) if ek == EK_F32 else to16(x, ek)


def dec(u, ek):
    return u.astype(np.float32) if ek == EK_F32 else from16(u, ek)


def act_layout(x_nchw_enc):
    """dd_elem.h: C >= 32 channel-blocked [B][C/32][h][w][32], else plain NHWC"""
    B, C, h, w = x_nchw_enc.shape
    if C >= 32:
        return blocked(x_nchw_enc)
    return np.ascontiguousarray(x_nchw_enc.transpose(0, 2, 3, 1))


def act_unlayout(buf, C):
    if C >= 32:
        return unblocked(buf, C)
    return np.ascontiguousarray(buf.transpose(0, 3, 1, 2))


def pack_weights(w_oihw, geom, ek):
    """dd_api.cpp pack_conv_weights (v2, swizzled): [n_tile][cin_chunk][tap_group][tap][n (NT)][k (CK)], zero beyond COUT, the 16-byte piece
    index of row r = tap * NT + n XORed with (r / (256 / rowbytes)) & (rowbytes / 16 - 1)"""
    cin, cout, cout_pad, ck, tg, nt, th, ks = geom
    e = esz(ek)
    wp = np.zeros((cout_pad, cin, ks * ks), np.float32)
    wp[:cout] = w_oihw.reshape(cout, cin, ks * ks)
    n_tiles, n_chunks, n_tg = cout_pad // nt, cin // ck, ks * ks // tg
    a = wp.reshape(n_tiles, nt, n_chunks, ck, n_tg, tg).transpose(0, 2, 4, 5, 1, 3)          # [tile][chunk][tg][t][n][k]
    a = np.ascontiguousarray(a).reshape(n_tiles, n_chunks, n_tg, tg * nt, ck)
    rowb = ck * e
    ppp, rpb, epp = rowb // 16, 256 // rowb, 16 // e
    rows = np.arange(tg * nt)
    out = np.empty_like(a).reshape(n_tiles, n_chunks, n_tg, tg * nt, ppp, epp)
    src = a.reshape(n_tiles, n_chunks, n_tg, tg * nt, ppp, epp)
    for piece in range(ppp):
        dst_piece = piece ^ ((rows // rpb) & (ppp - 1))
        out[:, :, :, rows, dst_piece] = src[:, :, :, rows, piece]
    return enc(out.reshape(-1), ek)


def gn_stats_slots(xr, B):
    """(sum, sum of squares) per GroupNorm group, spread over the 32 slots as the producing kernel leaves them"""
    st = np.zeros((B, STAT_SLOTS, STAT_STRIDE), np.float64)
    xg = xr.astype(np.float64).reshape(B, GN_GROUPS, -1)
    for g in range(GN_GROUPS):
        for sl, part in enumerate(np.array_split(xg[:, g], STAT_SLOTS, axis=1)):
            st[:, sl, 2 * g] = part.sum(1)
            st[:, sl, 2 * g + 1] = (part * part).sum(1)
    return st, xg


def gn_table(xg, gamma, beta, C):
    cg = C // GN_GROUPS
    mean, var = xg.mean(2), xg.var(2)
    a = gamma[None].astype(np.float64) / np.sqrt(np.repeat(var, cg, 1) + 1e-5)
    b = beta[None].astype(np.float64) - np.repeat(mean, cg, 1) * a
    return a.astype(np.float32), b.astype(np.float32)


# layer: (prologue, statistics follow, output fp32, relu epilogue)          dd_igemm2_cfg.h
SPEC = {
    1: ("x", True, False, False), 2: ("gn", True, False, False), 3: ("gn_add", True, False, False), 4: ("gn", True, True, False),
    5: ("gn_add", False, False, False), 6: ("raw", False, False, False), 7: ("raw", True, False, False),
    10: ("raw", False, False, True), 24: ("raw", False, False, True), 16: ("raw", False, False, True),
    14: ("raw", False, False, True),      # conv_up: ConvTranspose2d(k2, s2) as a 1x1 conv with 4 x 256 couts scattered to the output parities
    20: ("raw", False, False, False), 21: ("raw", False, False, False), 22: ("raw", False, False, False), 23: ("raw", False, True, False),
}


def run_layer(emu, layer, ek, *, B=1, h=11, w=37, order=0, late=0, seed=0, step=1, addend=False):
    g8 = (ctypes.c_int * 8)()
    emu.emu_geom2(layer, ek, g8)
    geom = tuple(g8)
    cin, cout, cout_pad, ck, tg, nt, th, ks = geom
    pro, stats, out_f32, relu = SPEC[layer]
    rng = np.random.default_rng(seed + 31 * layer)
    wt = (rng.standard_normal((cout, cin, ks, ks)) / np.sqrt(ks * ks * cin)).astype(np.float32)
    bias = np.zeros(cout_pad, np.float32)
    bias[:cout] = rng.standard_normal(cout).astype(np.float32) * 0.1
    wq = dec(enc(wt, ek), ek).astype(np.float64)                                          # the weights as the kernel sees them
    args = dict(in_=None, cond=None, emb=None, tvec=None, y4=None, xout=None, c1c2=None, stats_in=None, gamma=None, beta=None, addend=None)
    xout_ref = None
    if pro == "x":
        # conv1: x <- c1 x + c2 relu(gn4(y4)) fused into the load (step > 0), fp32 NHWC state, written back to xout
        x = rng.standard_normal((B, 16, h, w)).astype(np.float32)
        y4 = rng.standard_normal((B, 16, h, w)).astype(np.float32)
        gamma = (1.0 + 0.2 * rng.standard_normal(16)).astype(np.float32)
        beta = (0.3 * rng.standard_normal(16)).astype(np.float32)
        st, yg = gn_stats_slots(y4, B)
        a, b = gn_table(yg, gamma, beta, 16)
        c1c2 = rng.uniform(0.5, 1.2, size=(20, 2)).astype(np.float32)
        xin = x
        if step > 0:
            eps = np.maximum(a[:, :, None, None] * y4 + b[:, :, None, None], 0.0)
            xin = (c1c2[step - 1, 0] * x + c1c2[step - 1, 1] * eps).astype(np.float32)
            xout_ref = xin
        a_in = dec(enc(xin, ek), ek).astype(np.float64)                                  # operands are rounded to the MFMA type
        args.update(in_=np.ascontiguousarray(x.transpose(0, 2, 3, 1)), y4=np.ascontiguousarray(y4.transpose(0, 2, 3, 1)),
                    xout=np.full((B, h, w, 16), np.nan, np.float32), c1c2=c1c2, stats_in=st, gamma=gamma, beta=beta)
    else:
        x = rng.standard_normal((B, cin, h, w)).astype(np.float32)
        xe = enc(x, ek)
        xr = dec(xe, ek)
        args["in_"] = act_layout(xe)
        a_in = xr.astype(np.float64)
        if pro in ("gn", "gn_add"):
            gamma = (1.0 + 0.2 * rng.standard_normal(cin)).astype(np.float32)
            beta = (0.3 * rng.standard_normal(cin)).astype(np.float32)
            st, xg = gn_stats_slots(xr, B)
            a, b = gn_table(xg, gamma, beta, cin)
            v = np.maximum(a[:, :, None, None] * xr + b[:, :, None, None], 0.0).astype(np.float32)
            args.update(stats_in=st, gamma=gamma, beta=beta)
            if pro == "gn_add":
                emb = (0.2 * rng.standard_normal((1280, 256))).astype(np.float32)
                tvec = rng.integers(0, 1280, size=B + 3).astype(np.int64)
                ce = enc(rng.standard_normal((B, cin, h, w)).astype(np.float32), ek)
                t = tvec[2 + (np.arange(B) if B > 1 else np.zeros(B, np.int64))]
                v = v + (dec(ce, ek) + emb[t][:, :, None, None])
                args.update(cond=act_layout(ce), emb=emb, tvec=tvec)
            a_in = dec(enc(v, ek), ek).astype(np.float64)
    add_ref = None
    if addend:
        ae = enc(rng.standard_normal((B, cout, h, w)).astype(np.float32), ek)
        args["addend"] = act_layout(ae)
        add_ref = dec(ae, ek).astype(np.float64)
    wpack = pack_weights(wt, geom, ek)
    if out_f32:
        out = np.full((B, h, w, cout), np.nan, np.float32)
    elif layer == 14:
        out = np.full((B, 8, 2 * h, 2 * w, 32), 0x7E00 if ek == EK_F16 else 0x7FC0, np.uint16)
    elif ek == EK_F32:
        out = np.full((B, cout // 32, h, w, 32), np.nan, np.float32)
    else:
        out = np.full((B, cout // 32, h, w, 32), 0x7E00 if ek == EK_F16 else 0x7FC0, np.uint16)
    st_out = np.zeros((B, STAT_SLOTS, STAT_STRIDE), np.float64)
    emu.emu_set_order(order)
    emu.emu_set_dma_late(late)
    rc = emu.emu_conv2(layer, ek, ptr(args["in_"]), ptr(wpack), ptr(bias), ptr(out), ptr(st_out), ptr(args["stats_in"]), ptr(args["gamma"]),
                       ptr(args["beta"]), ptr(args["cond"]), ptr(args["emb"]), ptr(args["tvec"]), 2, 1 if B > 1 else 0, ptr(args["y4"]),
                       ptr(args["xout"]), ptr(args["c1c2"]), step if pro == "x" else 0, B, h, w, None, None, ptr(args["addend"]))
    assert rc == 0
    if out_f32:
        got = np.ascontiguousarray(out.transpose(0, 3, 1, 2)).astype(np.float64)
    elif layer == 14:
        up = dec(act_unlayout(out, 256), ek).astype(np.float64)                           # (B, 256, 2h, 2w)
        got = np.stack([up[:, :, dy::2, dx::2] for dy in (0, 1) for dx in (0, 1)], axis=1).reshape(B, 1024, h, w)   # cout block = parity
    else:
        got = dec(act_unlayout(out, cout), ek).astype(np.float64)
    ref = F.conv2d(torch.from_numpy(a_in), torch.from_numpy(wq), torch.from_numpy(bias[:cout].astype(np.float64)), padding=ks // 2).numpy()
    if relu:
        ref = np.maximum(ref, 0.0)
    if add_ref is not None:
        ref = ref + add_ref
    assert np.isfinite(got).all(), "unwritten output pixels"
    err = np.sqrt(np.mean((got - ref) ** 2)) / np.sqrt(np.mean(ref ** 2))
    tol = {EK_F32: 2e-6, EK_F16: 6e-4, EK_BF16: 5e-3}[ek] if not out_f32 else {EK_F32: 2e-6, EK_F16: 2e-5, EK_BF16: 2e-5}[ek]
    assert err < tol, (layer, ek, err)
    if xout_ref is not None:
        np.testing.assert_allclose(args["xout"].transpose(0, 3, 1, 2), xout_ref, rtol=2e-6, atol=2e-6)
    if stats:
        s = st_out.sum(1)
        gg = got.reshape(B, GN_GROUPS, -1)
        k = 1e-5 if (out_f32 or ek == EK_F32) else (1.5e-3 if ek == EK_F16 else 1.2e-2)
        np.testing.assert_allclose(s[:, 0:8:2], gg.sum(2), rtol=0, atol=k * np.sqrt((gg ** 2).sum(2)).max() + 1e-4)
        np.testing.assert_allclose(s[:, 1:8:2], (gg ** 2).sum(2), rtol=max(k, 2e-5))
    else:
        assert not st_out.any()
    return got


MODES = [(0, 0), (1, 0), (0, 1), (1, 1)]       # (which wave runs ahead, DMA lands at issue / as late as the waits allow)


@pytest.mark.parametrize("layer", [1, 2, 3, 4])
def test_res_denoiser_layers_all_timings(emu, layer):
    """conv1..conv4 of the headline loop, f16: identical results under the four adversarial timings"""
    outs = [run_layer(emu, layer, EK_F16, order=o, late=l) for o, l in MODES]
    for o in outs[1:]:
        np.testing.assert_array_equal(o, outs[0])


@pytest.mark.parametrize("layer", [5, 6, 7])
def test_swin_denoiser_layers_all_timings(emu, layer):
    outs = [run_layer(emu, layer, EK_F16, order=o, late=l) for o, l in MODES]
    for o in outs[1:]:
        np.testing.assert_array_equal(o, outs[0])


@pytest.mark.parametrize("layer,ek", [(1, EK_BF16), (2, EK_BF16), (3, EK_BF16), (4, EK_BF16), (3, EK_F32), (2, EK_F32), (7, EK_BF16)])
def test_other_element_kinds(emu, layer, ek):
    run_layer(emu, layer, ek, order=1, late=1)


def test_conv1_first_step_without_update(emu):
    """step 0: the state is used as stored, nothing is written back"""
    run_layer(emu, 1, EK_F16, step=0, late=1)


def test_fpn_upsampling_layer(emu):
    run_layer(emu, 14, EK_F16, order=1, late=1, h=9, w=35)


@pytest.mark.parametrize("layer", [10, 24, 16])
def test_fpn_lateral_layers(emu, layer):
    """relu(conv + folded BN) + top-down term; ResNet 64, MPViT 128, Swin-L 384 input channels"""
    run_layer(emu, layer, EK_F16, order=1, late=1, addend=True)
    run_layer(emu, layer, EK_F16, order=0, late=0, addend=False)


@pytest.mark.parametrize("layer", [20, 21, 22, 23])
def test_data_gradient_layers(emu, layer):
    """backward of conv4 .. conv1: the same implicit GEMM on transposed / flipped weights (the transposition is the host's, not tested here)"""
    run_layer(emu, layer, EK_F16, order=1, late=1)


def test_batch_two_images(emu):
    run_layer(emu, 3, EK_F16, B=2, h=9, w=33, order=1, late=1)


def test_emulation_catches_a_lax_counted_wait(emu, tmp_path):
    """The checker's own sensitivity (a removed loop barrier turns the same cases red: that mutant lived in round 1's Winograd harness, deleted with those kernels): the stage's
    `s_waitcnt vmcnt(NRAW)` leaves exactly the NRAW raw-patch loads in flight and so retires the weight DMA issued before them.  With
    NRAW + 1 the wave's last DMA piece may still be in flight at the barrier: only the late-landing model can see that, and it must."""
    old = "      DD_WAIT_VM_LGKM0(NRAW);\n    } else {"
    mut = bind_igemm2(build_mutant("dd_igemm2.hip", old, "      DD_WAIT_VM_LGKM0(NRAW + 1);\n    } else {", tmp_path, count=1))
    # layer 7 (Swin pred.0, 256 -> 64 on a raw input): two raw-patch register slots, i.e. the counted wait is on its path
    run_layer(mut, 7, EK_F16, order=0, late=0)              # DMA lands at issue: the lax wait is invisible
    with pytest.raises(AssertionError):
        run_layer(mut, 7, EK_F16, order=0, late=1)          # DMA lands as late as the waits allow: stale weights
