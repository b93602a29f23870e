"""The Winograd kernels of diffusiondepth_amd/csrc/dd_wino.hip (experimental, option "winograd"; all but v1 have never had GPU time) EXECUTED on
the CPU: the kernel source itself is compiled for the host on top of tests/host_emul/hip/hip_runtime.h (work-items = fibers, LDS = an array,
MFMA / shuffle = wave-level exchanges in the documented register layout) and run workgroup by workgroup, then compared with an fp64
convolution of the same operands.  What this pins before the first GPU contact: the index arithmetic of every LDS image (raw patch with its
column swizzle, U / V with their k-half swizzle), the double-buffer and barrier scheme (a work-item that runs ahead as far as the barriers allow
must not overwrite anything another one still needs: both schedule orders), the prologue table, the GroupNorm statistics epilogue, the
LDS-DMA source-side swizzle (DMA = copy at issue), ragged tiles, and the launchers' template dispatch.  What it cannot pin: speed, bank
conflicts, and the hardware's own timing of asynchronous copies.

Test infrastructure only (like oracle/): nothing in the product path uses the host build."""
from __future__ import annotations


import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hostemu_util import EK_BF16, EK_F16, GN_GROUPS, STAT_SLOTS, STAT_STRIDE, bind_wino, blocked, build_library, build_mutant, from16, ptr, to16, unblocked


@pytest.fixture(scope="module")
def emu():
    return bind_wino(build_library())


LAYERS = {  # layer: (cin, cout, prologue, statistics)     dd_wino.hip launch_wino_layer_ek
    2: (64, 256, "gn", True),
    3: (256, 64, "gn_add", True),
    5: (256, 256, "gn_add", False),
    6: (256, 256, "raw", False),
    7: (256, 64, "raw", True),
}


def run_case(emu, layer, ek, *, version=2, packed=False, dma=False, B=1, h=11, w=37, order=0, seed=0):
    cin, cout, pro, stats = LAYERS[layer]
    rng = np.random.default_rng(seed + 17 * layer)
    x = rng.standard_normal((B, cin, h, w)).astype(np.float32)
    x16 = to16(x, ek)
    xr = from16(x16, ek)
    wt = (rng.standard_normal((cout, cin, 3, 3)) / np.sqrt(9 * cin)).astype(np.float32)
    bias = rng.standard_normal(cout).astype(np.float32) * 0.1
    cond16 = tab = None
    a_in = xr.astype(np.float64)
    if pro != "raw":
        # the producer's GroupNorm partial sums, spread over the 32 slots as the kernels leave them
        gamma = (1.0 + 0.2 * rng.standard_normal(cin)).astype(np.float32)
        beta = (0.3 * rng.standard_normal(cin)).astype(np.float32)
        cg = cin // GN_GROUPS
        st = np.zeros((B, STAT_SLOTS, STAT_STRIDE), np.float64)
        xg = xr.astype(np.float64).reshape(B, GN_GROUPS, -1)
        for g in range(GN_GROUPS):
            parts = np.array_split(xg[:, g], STAT_SLOTS, axis=1)
            for sl, part in enumerate(parts):
                st[:, sl, 2 * g] = part.sum(1)
                st[:, sl, 2 * g + 1] = (part * part).sum(1)
        emb = tvec = None
        if pro == "gn_add":
            emb = (0.2 * rng.standard_normal((1280, 256))).astype(np.float32)
            tvec = rng.integers(0, 1280, size=B + 3).astype(np.int64)
            cond = rng.standard_normal((B, cin, h, w)).astype(np.float32)
            cond16 = to16(cond, ek)
        tab = np.zeros((B, cin, 4), np.float32)
        rc = emu.emu_wino_table(ptr(st), ptr(gamma), ptr(beta), ptr(emb), ptr(tvec), 2, 1 if B > 1 else 0, B, h, w, cin, ptr(tab))
        assert rc == 0
        # the table against its definition
        mean = xg.mean(2)
        var = xg.var(2)
        a_ref = gamma[None] / np.sqrt(np.repeat(var, cg, 1) + 1e-5)
        b_ref = beta[None] - np.repeat(mean, cg, 1) * a_ref
        np.testing.assert_allclose(tab[:, :, 0], a_ref, rtol=2e-6, atol=1e-6)
        np.testing.assert_allclose(tab[:, :, 1], b_ref, rtol=2e-5, atol=2e-6)
        v = np.maximum(tab[:, :, 0, None, None] * xr + tab[:, :, 1, None, None], 0.0).astype(np.float32)     # fp32 fma ~ fp32 mul + add here
        if pro == "gn_add":
            t = tvec[2 + (np.arange(B) if B > 1 else np.zeros(B, np.int64))]
            np.testing.assert_array_equal(tab[:, :, 2], emb[t])
            v = v + (from16(cond16, ek) + tab[:, :, 2, None, None])
        a_in = from16(to16(v, ek), ek).astype(np.float64)        # the prologue's result is rounded once to the operand type
    nbytes = emu.emu_wino_pack_bytes(cout, cin)
    wpack = np.zeros(nbytes // 2, np.uint16)
    emu.emu_wino_pack(ptr(wt), cout, cin, ek, ptr(wpack))
    out = np.full((B, cout // 32, h, w, 32), 0x7E00 if ek == EK_F16 else 0x7FC0, np.uint16)       # NaN: every pixel must be written
    st_out = np.zeros((B, STAT_SLOTS, STAT_STRIDE), np.float64)
    in_blk = blocked(x16)
    cond_blk = blocked(cond16) if cond16 is not None else None
    emu.emu_set_order(order)
    rc = emu.emu_wino_layer(version, layer, ek, int(packed), int(dma), ptr(in_blk), ptr(cond_blk), ptr(wpack), ptr(bias), ptr(tab), ptr(out),
                            ptr(st_out), B, h, w)
    assert rc == 0
    got = from16(unblocked(out, cout), ek).astype(np.float64)
    ref = F.conv2d(torch.from_numpy(a_in), torch.from_numpy(wt.astype(np.float64)), torch.from_numpy(bias.astype(np.float64)), padding=1).numpy()
    assert np.isfinite(got).all(), "unwritten output pixels"
    err = np.sqrt(np.mean((got - ref) ** 2)) / np.sqrt(np.mean(ref ** 2))
    worst = np.abs(got - ref).max() / np.abs(ref).max()
    if stats:
        # the statistics are taken from the kernel's own fp32 results (before the 16-bit rounding): compare with the sums over what it stored --
        # only the independent final roundings differ (2^-11 / 2^-8 relative per element); a pixel counted twice or not at all would be O(1)
        s = st_out.sum(1)
        gg = got.reshape(B, GN_GROUPS, -1)
        k = 1.5e-3 if ek == EK_F16 else 1.2e-2     # ~4 sigma of sqrt(N) independent roundings
        np.testing.assert_allclose(s[:, 0:8:2], gg.sum(2), rtol=0, atol=k * np.sqrt((gg ** 2).sum(2)).max())
        np.testing.assert_allclose(s[:, 1:8:2], (gg ** 2).sum(2), rtol=k)
        assert not s[:, 8:].any()
    else:
        assert not st_out.any()
    return err, worst


# relative rms error budget: Winograd with operands rounded to the 16-bit type (tools/winograd_numerics.py: ~2.3x the direct kernel's error)
TOL = {EK_F16: (1.5e-3, 1e-2), EK_BF16: (1.2e-2, 8e-2)}


@pytest.mark.parametrize("layer", [2, 3, 5, 6, 7])
def test_v2_layers_f16(emu, layer):
    err, worst = run_case(emu, layer, EK_F16)
    assert err < TOL[EK_F16][0] and worst < TOL[EK_F16][1], (err, worst)


def test_v2_runahead_last_workitem(emu):
    """the opposite schedule: the LAST work-item runs as far ahead as the barriers allow"""
    err, worst = run_case(emu, 3, EK_F16, order=1)
    assert err < TOL[EK_F16][0] and worst < TOL[EK_F16][1], (err, worst)


@pytest.mark.parametrize("layer", [2, 3])
def test_v2_bf16(emu, layer):
    err, worst = run_case(emu, layer, EK_BF16)
    assert err < TOL[EK_BF16][0] and worst < TOL[EK_BF16][1], (err, worst)


@pytest.mark.parametrize("layer", [3, 6])
def test_v2_packed_f16_transform(emu, layer):
    err, worst = run_case(emu, layer, EK_F16, packed=True)
    assert err < 2 * TOL[EK_F16][0] and worst < 2 * TOL[EK_F16][1], (err, worst)


@pytest.mark.parametrize("layer,order,late", [(2, 0, 0), (6, 1, 1), (3, 0, 1)])
def test_v2_dma_weight_images(emu, layer, order, late):
    """option "winograd_dma": same results as the register path bit for bit (the swizzle moves to the source side of the copy), whether the
    DMA lands at issue or only at the s_waitcnt in front of barrier B (dd_gcn.h: the two extremes of the hardware's timing)"""
    e0 = run_case(emu, layer, EK_F16, dma=False, order=order)
    emu.emu_set_dma_late(late)
    try:
        e1 = run_case(emu, layer, EK_F16, dma=True, order=order)
    finally:
        emu.emu_set_dma_late(0)
    assert e0 == e1


def test_v2_batch_and_ragged(emu):
    """two images (per-image table / timestep / statistics rows), a size whose last tiles hang over both edges"""
    err, worst = run_case(emu, 3, EK_F16, B=2, h=9, w=33)
    assert err < TOL[EK_F16][0] and worst < TOL[EK_F16][1], (err, worst)


def test_v1_matches_gpu_run(emu):
    """v1 (convB) has run on the GPU (run 40: eps error 1.09x the direct kernel's): the emulation must accept it too -- the check of the checker"""
    err, worst = run_case(emu, 6, EK_F16, version=1)
    assert err < TOL[EK_F16][0] and worst < TOL[EK_F16][1], (err, worst)


# ---- the checker's own sensitivity: kernels with one synchronisation point or one swizzle removed must FAIL --------------------------------------
MUTATIONS = {
    "no barrier A (raw / U of the next chunk published)":
        ("    __syncthreads();                                            // A: raw (and, without DMA, U) of chunk + 1 visible\n", "\n"),
    "no barrier B (V of the next chunk complete, fragment reads done)":
        ("    __syncthreads();                                            // B: V (DMA: and U) of chunk + 1 complete, fragment reads of this chunk done\n", "\n"),
    "fragment reads without the k-half swizzle":
        ("((g ^ ((li >> 3) & 1)) * 8)) * 2);", "(g * 8)) * 2);"),
    "raw image written without the column swizzle":
        ("    return (pr * W_PW + (pc ^ ((pc >> 2) & 1))) * (W_CK * 2) + (item & 1) * 16;", "    return (pr * W_PW + pc) * (W_CK * 2) + (item & 1) * 16;"),
}


@pytest.mark.parametrize("name", list(MUTATIONS))
def test_emulation_catches_mutation(emu, name, tmp_path):
    """without this the green cases above would prove little: the same cases must go red when the kernel is broken in the ways the emulation
    is meant to catch (a wave that runs ahead past a missing barrier; an LDS image read differently from how it was written)"""
    old, new = MUTATIONS[name]
    mut = bind_wino(build_mutant("dd_wino.hip", old, new, tmp_path))
    bad = 0
    for order in (0, 1):
        try:
            err, worst = run_case(mut, 6, EK_F16, order=order)
            bad += int(not (err < TOL[EK_F16][0] and worst < TOL[EK_F16][1]))
        except AssertionError:
            bad += 1
    assert bad >= 1, "the emulation did not notice: " + name
