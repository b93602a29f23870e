"""CPU: statement-by-statement NumPy emulation of ONE WORKGROUP of the experimental Winograd kernel (csrc/dd_wino.hip,
conv_wino_raw_kernel) -- its LDS layouts, lane roles, MFMA operand / accumulator lane mapping, position-to-wave assignment, epilogue
gather and global addressing -- fed with the weight image the library itself packs (dd_debug_wino_pack, host-only), against
F.conv2d.  The kernel was written in round 1 with no GPU time left to run it; this test pins everything about it that can be pinned
without a GPU.  Assumed from the validated direct kernels (dd_elem.h mma_step / dd_igemm2.hip epilogue): v_mfma_f32_32x32x16 with
A = weights (row = lane % 32, k = 8 * (lane / 32) .. +7), B = tiles (column = lane % 32, same k), D register r of a lane =
row 8 * (r / 4) + 4 * (lane / 32) + r % 4, column lane % 32."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import diffusiondepth_amd as dda

TH, TW, TX, TILES, PH, PW, CK, NT = 8, 32, 16, 64, 10, 34, 16, 64
C = 256


def _f16(x):
    return np.asarray(x, np.float32).astype(np.float16)


def _act_offset(Cc, h, w, b, c, y, x):                       # dd_elem.h act_offset
    return ((((b * (Cc // 32) + c // 32) * h + y) * w + x) * 32) + c % 32


def _emulate_workgroup(inp_blocked, upack, bias, h, w, b, ty0, tx0, nsplit, out_blocked, C=256, COUT=256, tab=None, cond_blocked=None,
                       v2_epilogue=False):
    """inp_blocked / out_blocked: flat f16 arrays in the activation layout [B][C/32][h][w][32]; upack: flat uint16 image.
    tab (generalised kernel): [B][C][4] floats (a, b, e, -) of the GroupNorm (+ condition) prologue applied when a piece goes to LDS;
    cond_blocked: the condition map in the input's layout.  Returns the workgroup's GroupNorm partial sums {slot: [sum, sumsq]}."""
    HW = h * w
    y0, x0 = ty0 * TH, tx0 * TW
    acc = np.zeros((8, 2, 2, 2, 64, 16), np.float32)          # [wave][a][m][n][lane][r]
    lane = np.arange(64)
    li, g = lane & 31, lane >> 5
    stats = {}
    for chunk in range(C // CK):
        # (a) raw patch and U chunk -> "LDS" (the generalised kernel applies the prologue to each 8-channel piece on the way)
        s_raw = np.zeros((PH * PW, CK), np.float16)
        cbase = (chunk >> 1) * HW * 32 + (chunk & 1) * CK
        swz = (lambda c: c ^ ((c >> 2) & 1)) if v2_epilogue else (lambda c: c)     # v2: bank-conflict-free column order of the raw image
        for pp_lin in range(PH * PW):
            pr, pc = divmod(pp_lin, PW)
            pp = pr * PW + swz(pc)
            gy, gx = y0 - 1 + pr, x0 - 1 + pc
            if 0 <= gy < h and 0 <= gx < w:
                o = b * HW * C + cbase + (gy * w + gx) * 32
                for hf in range(2):
                    v = inp_blocked[o + hf * 8:o + hf * 8 + 8].astype(np.float32)
                    if tab is not None:
                        ch0 = chunk * CK + hf * 8
                        t = tab[b, ch0:ch0 + 8]
                        v = np.maximum(t[:, 0] * v + t[:, 1], np.float32(0))
                        if cond_blocked is not None:
                            v = v + (cond_blocked[o + hf * 8:o + hf * 8 + 8].astype(np.float32) + t[:, 2])
                    s_raw[pp, hf * 8:hf * 8 + 8] = v.astype(np.float16)
        u0 = (nsplit * (C // CK) + chunk) * 16 * NT * CK
        s_u = upack[u0:u0 + 16 * NT * CK].view(np.float16).reshape(16, NT, CK)
        if v2_epilogue:
            # v2 stores piece q = (row, half) of the packed image at q ^ ((row >> 3) & 1): rows with bit 3 set hold their k-halves swapped
            phys = np.zeros_like(s_u)
            for row in range(NT):
                f = (row >> 3) & 1
                phys[:, row, :8], phys[:, row, 8:] = (s_u[:, row, 8:], s_u[:, row, :8]) if f else (s_u[:, row, :8], s_u[:, row, 8:])
            s_u = phys
        hsel = (lambda l_i, g_: (g_ ^ ((l_i >> 3) & 1)) * 8) if v2_epilogue else (lambda l_i, g_: g_ * 8)   # where a lane finds its k-half
        # (b) transform: thread tid -> tile tt = tid >> 3, channel pair cp = tid & 7
        s_v = np.zeros((16, TILES, CK), np.float16)
        for tid in range(512):
            cp, tt = tid & 7, tid >> 3
            tty, ttx = divmod(tt, TX)
            d = np.zeros((2, 4, 4), np.float32)
            for i in range(4):
                for j in range(4):
                    pk = s_raw[(2 * tty + i) * PW + swz(2 * ttx + j), 2 * cp:2 * cp + 2].astype(np.float32)
                    d[0, i, j], d[1, i, j] = pk[0], pk[1]
            for c in range(2):
                t = np.stack([d[c, 0] - d[c, 2], d[c, 1] + d[c, 2], d[c, 2] - d[c, 1], d[c, 1] - d[c, 3]])
                v = np.stack([t[:, 0] - t[:, 2], t[:, 1] + t[:, 2], t[:, 2] - t[:, 1], t[:, 1] - t[:, 3]], axis=1)
                for i in range(4):
                    for j in range(4):
                        chv = (2 * cp + c) ^ ((((tt >> 3) & 1) << 3) if v2_epilogue else 0)
                        s_v[i * 4 + j, tt, chv] = np.float16(v[i, j])
        # (c) MFMA: wave q, positions 2q + a; fragments per the lane mapping
        for wave in range(8):
            for a in range(2):
                pos = 2 * wave + a
                for m in range(2):
                    A = np.zeros((32, 16), np.float32)
                    for l in range(64):
                        A[li[l], g[l] * 8:g[l] * 8 + 8] = s_u[pos, m * 32 + li[l], hsel(li[l], g[l]):hsel(li[l], g[l]) + 8]
                    for n in range(2):
                        Bm = np.zeros((16, 32), np.float32)
                        for l in range(64):
                            Bm[g[l] * 8:g[l] * 8 + 8, li[l]] = s_v[pos, n * 32 + li[l], hsel(li[l], g[l]):hsel(li[l], g[l]) + 8]
                        D = A @ Bm                                  # 32 (cout) x 32 (tile)
                        for r in range(16):
                            rows = 8 * (r // 4) + 4 * g + r % 4
                            acc[wave, a, m, n, lane, r] += D[rows, li]
    # epilogue
    for blk in range(4):
        m, n = blk >> 1, blk & 1
        s_m = np.zeros((16, 32, 32), np.float32)                # [pos][tile][co]
        for wave in range(8):
            for a in range(2):
                pos = 2 * wave + a
                for q in range(4):
                    for l in range(64):
                        s_m[pos, li[l], 8 * q + 4 * g[l]:8 * q + 4 * g[l] + 4] = acc[wave, a, m, n, l, 4 * q:4 * q + 4]
        if v2_epilogue:
            # the double-buffered kernel: all 512 lanes, lane -> (tile, cout quad, output row dy)
            for tid in range(512):
                tj, cg, c8, dy = tid >> 4, (tid >> 2) & 3, (tid >> 1) & 1, tid & 1
                T = n * 32 + tj
                ty, tx = divmod(T, TX)
                co = nsplit * NT + m * 32 + cg * 8 + c8 * 4
                cs = slice(cg * 8 + c8 * 4, cg * 8 + c8 * 4 + 4)
                t = np.zeros((4, 4), np.float32)
                for j in range(4):
                    r1, r2, re = s_m[1 * 4 + j, tj, cs], s_m[2 * 4 + j, tj, cs], s_m[(3 if dy else 0) * 4 + j, tj, cs]
                    t[j] = (re + r1 + r2) if dy == 0 else (r1 - r2 - re)
                gy = y0 + 2 * ty + dy
                for dx in range(2):
                    gx = x0 + 2 * tx + dx
                    if gy < h and gx < w:
                        v = ((t[0] + t[1] + t[2]) if dx == 0 else (t[1] - t[2] - t[3])) + bias[co:co + 4]
                        o = b * HW * COUT + _act_offset(COUT, h, w, 0, co, gy, gx)
                        out_blocked[o:o + 4] = v.astype(np.float16)
                        slot = 0 if COUT == 256 else (m * 32 + cg * 8) >> 4
                        st = stats.setdefault((nsplit if COUT == 256 else 0) + slot, np.zeros(2, np.float64))
                        st += [float(v.astype(np.float64).sum()), float((v.astype(np.float64) ** 2).sum())]
            continue
        for tid in range(128):
            tj, cg = tid >> 2, tid & 3
            T = n * 32 + tj
            ty, tx = divmod(T, TX)
            co = nsplit * NT + m * 32 + cg * 8
            for c8 in range(2):
                mm = s_m[:, tj, cg * 8 + c8 * 4:cg * 8 + c8 * 4 + 4].reshape(4, 4, 4)      # [i][j][c]
                t0 = mm[0] + mm[1] + mm[2]
                t1 = mm[1] - mm[2] - mm[3]
                for dy in range(2):
                    tr = t0 if dy == 0 else t1
                    for dx in range(2):
                        gy, gx = y0 + 2 * ty + dy, x0 + 2 * tx + dx
                        if gy < h and gx < w:
                            v = (tr[0] + tr[1] + tr[2]) if dx == 0 else (tr[1] - tr[2] - tr[3])
                            v = v + bias[co + c8 * 4:co + c8 * 4 + 4]
                            o = b * HW * COUT + _act_offset(COUT, h, w, 0, co + c8 * 4, gy, gx)
                            out_blocked[o:o + 4] = v.astype(np.float16)
                            slot = 0 if COUT == 256 else (m * 32 + cg * 8) >> 4            # local GroupNorm slot, as the kernel
                            gslot = (nsplit if COUT == 256 else 0) + slot
                            st = stats.setdefault(gslot, np.zeros(2, np.float64))
                            st += [float(v.astype(np.float64).sum()), float((v.astype(np.float64) ** 2).sum())]
    return stats


def _to_blocked(x):                                          # (B,C,h,w) -> flat [B][C/32][h][w][32]
    B, Cc, h, w = x.shape
    return np.ascontiguousarray(x.reshape(B, Cc // 32, 32, h, w).transpose(0, 1, 3, 4, 2)).reshape(-1)


def test_winograd_workgroup_emulation_matches_conv2d():
    lib = dda.load_library()
    rs = np.random.RandomState(0)
    B, h, w = 1, 11, 37                                      # ragged: partial tiles in both directions, odd sizes
    x = _f16(rs.standard_normal((B, C, h, w)))
    wgt = (rs.standard_normal((C, C, 3, 3)) / np.sqrt(C * 9)).astype(np.float32)
    bias = rs.standard_normal(C).astype(np.float32)
    up = np.zeros(C * C * 16, np.uint16)
    rc = lib.dd_debug_wino_pack(wgt.ctypes.data_as(ctypes.c_void_p), C, C, dda.precision_id("f16"), up.ctypes.data_as(ctypes.c_void_p), up.size)
    assert rc == 0
    # the packed image is G g G^T at [split][chunk][pos][co][ck]
    G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
    U = np.einsum("ij,ocjk,lk->ocil", G, wgt.astype(np.float64), G).astype(np.float32).astype(np.float16)
    img = up.view(np.float16).reshape(C // NT, C // CK, 16, NT, CK)
    for (sp, ch, pos, cl, ck) in ((0, 0, 0, 0, 0), (3, 15, 15, 63, 15), (1, 7, 6, 20, 9), (2, 3, 11, 5, 2)):
        assert img[sp, ch, pos, cl, ck] == U[sp * NT + cl, ch * CK + ck, pos // 4, pos % 4]
    ref = F.conv2d(torch.from_numpy(x.astype(np.float32)), torch.from_numpy(wgt), torch.from_numpy(bias), padding=1).numpy()
    inp_b = _to_blocked(x)
    out_b = np.zeros(B * C * h * w, np.float16)
    # two workgroups: the last (ragged) tile of the image with the last cout split, and the first tile with split 1
    for (ty0, tx0, nsplit) in (((h - 1) // TH, (w - 1) // TW, 3), (0, 0, 1)):
        _emulate_workgroup(inp_b, up, bias, h, w, 0, ty0, tx0, nsplit, out_b)
        got = out_b.reshape(B, C // 32, h, w, 32).transpose(0, 1, 4, 2, 3).reshape(B, C, h, w).astype(np.float32)
        ys, xs = slice(ty0 * TH, min(h, ty0 * TH + TH)), slice(tx0 * TW, min(w, tx0 * TW + TW))
        cs = slice(nsplit * NT, nsplit * NT + NT)
        err = np.abs(got[:, cs, ys, xs] - ref[:, cs, ys, xs]).max()
        assert err < 6e-3 * max(1.0, np.abs(ref).max()), (ty0, tx0, nsplit, err)       # f16 operands (V, U rounded) + f16 output


def test_generalised_winograd_kernel_emulation_conv3_like():
    """The generalised kernel as conv3 (256 -> 64, prologue relu(a*y + b) + cond + e from the table, GroupNorm partial sums out):
    prologue channel indexing, the [B][C][4] table, output layout of a 64-channel tensor and the statistics slots."""
    lib = dda.load_library()
    rs = np.random.RandomState(1)
    B, h, w, CIN, COUT = 2, 9, 35, 256, 64
    y2 = _f16(rs.standard_normal((B, CIN, h, w)))
    cond = _f16(np.abs(rs.standard_normal((B, CIN, h, w))))
    tab = np.zeros((B, CIN, 4), np.float32)
    tab[..., 0] = rs.uniform(0.5, 1.5, (B, CIN)); tab[..., 1] = rs.uniform(-0.3, 0.3, (B, CIN)); tab[..., 2] = rs.uniform(-0.5, 0.5, (B, CIN))
    wgt = (rs.standard_normal((COUT, CIN, 3, 3)) / np.sqrt(CIN * 9)).astype(np.float32)
    bias = rs.standard_normal(COUT).astype(np.float32)
    up = np.zeros(COUT * CIN * 16, np.uint16)
    assert lib.dd_debug_wino_pack(wgt.ctypes.data_as(ctypes.c_void_p), COUT, CIN, dda.precision_id("f16"), up.ctypes.data_as(ctypes.c_void_p), up.size) == 0
    bi = 1
    f = np.maximum(tab[bi, :, 0, None, None] * y2[bi].astype(np.float32) + tab[bi, :, 1, None, None], 0) + (cond[bi].astype(np.float32) + tab[bi, :, 2, None, None])
    f = f.astype(np.float16).astype(np.float32)                                  # the operand the kernels see
    ref = F.conv2d(torch.from_numpy(f[None]), torch.from_numpy(wgt), torch.from_numpy(bias), padding=1).numpy()[0]
    out_b = np.zeros(B * COUT * h * w, np.float16)
    ty0, tx0 = (h - 1) // TH, (w - 1) // TW                                      # the ragged corner tile
    stats = _emulate_workgroup(_to_blocked(y2), up, bias, h, w, bi, ty0, tx0, 0, out_b, C=CIN, COUT=COUT, tab=tab, cond_blocked=_to_blocked(cond),
                               v2_epilogue=True)
    got = out_b.reshape(B, COUT // 32, h, w, 32).transpose(0, 1, 4, 2, 3).reshape(B, COUT, h, w).astype(np.float32)[bi]
    ys, xs = slice(ty0 * TH, h), slice(tx0 * TW, w)
    assert np.abs(got[:, ys, xs] - ref[:, ys, xs]).max() < 6e-3 * max(1.0, np.abs(ref).max())
    assert np.abs(got[:, :ty0 * TH]).max() == 0.0                                # nothing written outside the tile
    for grp in range(4):                                                        # GroupNorm groups of 16 couts
        r = ref[grp * 16:(grp + 1) * 16, ys, xs].astype(np.float64)
        assert abs(stats[grp][0] - r.sum()) < 2e-2 * max(1.0, abs(r.sum())) and abs(stats[grp][1] - (r ** 2).sum()) < 1e-2 * (r ** 2).sum()


def test_generalised_winograd_kernel_emulation_conv2_like():
    """conv2 instance: 64 -> 256 (4 chunks, 4 cout splits), prologue relu(a*y + b) without a condition term, statistics of GroupNorm group
    `nsplit` (the workgroup's 64 couts are exactly one of the 4 groups of 64)."""
    lib = dda.load_library()
    rs = np.random.RandomState(2)
    B, h, w, CIN, COUT = 1, 8, 32, 64, 256
    y1 = _f16(rs.standard_normal((B, CIN, h, w)))
    tab = np.zeros((B, CIN, 4), np.float32)
    tab[..., 0] = rs.uniform(0.5, 1.5, (B, CIN)); tab[..., 1] = rs.uniform(-0.3, 0.3, (B, CIN))
    wgt = (rs.standard_normal((COUT, CIN, 3, 3)) / np.sqrt(CIN * 9)).astype(np.float32)
    bias = rs.standard_normal(COUT).astype(np.float32)
    up = np.zeros(COUT * CIN * 16, np.uint16)
    assert lib.dd_debug_wino_pack(wgt.ctypes.data_as(ctypes.c_void_p), COUT, CIN, dda.precision_id("f16"), up.ctypes.data_as(ctypes.c_void_p), up.size) == 0
    a1 = np.maximum(tab[0, :, 0, None, None] * y1[0].astype(np.float32) + tab[0, :, 1, None, None], 0).astype(np.float16).astype(np.float32)
    ref = F.conv2d(torch.from_numpy(a1[None]), torch.from_numpy(wgt), torch.from_numpy(bias), padding=1).numpy()[0]
    out_b = np.zeros(B * COUT * h * w, np.float16)
    nsplit = 2
    stats = _emulate_workgroup(_to_blocked(y1), up, bias, h, w, 0, 0, 0, nsplit, out_b, C=CIN, COUT=COUT, tab=tab, v2_epilogue=True)
    got = out_b.reshape(B, COUT // 32, h, w, 32).transpose(0, 1, 4, 2, 3).reshape(B, COUT, h, w).astype(np.float32)[0]
    cs = slice(nsplit * NT, nsplit * NT + NT)
    assert np.abs(got[cs] - ref[cs]).max() < 6e-3 * max(1.0, np.abs(ref).max())
    assert set(stats) == {nsplit}
    r = ref[cs].astype(np.float64)
    assert abs(stats[nsplit][0] - r.sum()) < 2e-2 * max(1.0, abs(r.sum())) and abs(stats[nsplit][1] - (r ** 2).sum()) < 1e-2 * (r ** 2).sum()


def test_winograd_lds_layouts_are_bank_conflict_free():
    """tools/lds_bank_check.py (rules of MI355X_MICROARCH.md section LDS): the swizzled LDS images of the double-buffered kernel have no
    bank conflicts in any of its four access patterns; the linear layouts would be 2-way for the transform reads and the fragment reads."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import lds_bank_check as L
    assert (L.transform_reads(True), L.v_writes(True), L.fragment_reads(True), L.raw_stores(True)) == (1, 1, 1, 1)
    assert L.transform_reads(False) == 2 and L.fragment_reads(False) == 2
