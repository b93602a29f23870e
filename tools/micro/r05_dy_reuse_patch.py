# Applies the row-reuse MFMA block of profiles/r05_experiments.md section 2 to dd_igemm2.hip / dd_igemm2_cfg.h IN PLACE (the sources of the commit it was cut from;
# every anchor is asserted).  Run it as `python tools/micro/r05_dy_reuse_patch.py apply` -- without the word it does nothing; undo with `git checkout -- diffusiondepth_amd/csrc`.
import sys
if sys.argv[1:] != ["apply"]:
    sys.exit("usage: python tools/micro/r05_dy_reuse_patch.py apply   (edits diffusiondepth_amd/csrc in place)")
p='/root/repo/diffusiondepth_amd/csrc/dd_igemm2.hip'
s=open(p).read()
old='''    int wa[NKQ];
#pragma unroll
    for (int kq = 0; kq < NKQ; ++kq) wa[kq] = wkt[kq] + woff;
    {
'''
new='''    int wa[NKQ];
#pragma unroll
    for (int kq = 0; kq < NKQ; ++kq) wa[kq] = wkt[kq] + woff;
    if constexpr (C::DYR) {
      // Column-major taps with ROW REUSE (round 5; conv3-shaped layers, nine taps of a 16-channel chunk per stage): for one column shift dx the wave
      // reads the WM + 2 patch rows it touches ONCE and uses each of them for up to three taps (dy = 0, 1, 2 shift the SAME fragments by one output
      // row): 3 (WM + 2) pixel + 9 WN weight fragment reads per stage instead of 9 (WM + WN) -- 30 instead of 36 per 36 MFMAs on 8x32 tiles
      // (WM = WN = 2), 36 instead of 54 per 72 on 16x32 tiles (WM = 4: 0.5 reads per MFMA).  Row r of the buffer is dead behind tap dy = min(r, 2) of
      // its column, so the next column's rows 0 and 1 are fetched under the taps dy = 1, 2 of this one and the rest under tap 0 of the next;
      // rows 2 .. WM - 1 (first used at dy = 0 AND last used at dy = 2: 16x32 tiles only) alternate between two registers.  Weight fragments
      // of the next tap are fetched under the MFMAs of this one.  The accumulation order inside a chunk becomes (dx, dy) instead of (dy, dx).
      static_assert(C::TG == 9 && NKQ == 1 && C::KS == 3 && !C::SPLIT && EK != EK_F32, "row-reuse form: nine taps per stage, one k-step per tap");
      constexpr int NP = C::WM + 2;
      uint4 pf[2][NP];
      uint4 wf[2][C::WN];
      auto prow = [&](int dx, int r) -> uint4& { return pf[(r >= 2 && r <= C::WM - 1) ? (dx & 1) : 0][r]; };
      auto load_p = [&](int dx, int r) { prow(dx, r) = *reinterpret_cast<const uint4*>(smem + colt[dx][0] + poff + r * (PW * ROWB)); };
      auto load_w = [&](int dx, int dy, uint4 (&f)[C::WN]) {
#pragma unroll
        for (int n = 0; n < C::WN; ++n) f[n] = *reinterpret_cast<const uint4*>(smem + wa[0] + ((dy * 3 + dx) * C::NT + n * 32) * ROWB);
      };
#pragma unroll
      for (int r = 0; r < C::WM; ++r) load_p(0, r);          // what tap (dx 0, dy 0) reads; the remaining rows follow under its MFMAs
      load_w(0, 0, wf[0]);
      auto tap = [&](auto DXc, auto DYc) {
        constexpr int dx = decltype(DXc)::value, dy = decltype(DYc)::value;
        constexpr int cur = (dx * 3 + dy) & 1;
        constexpr int NMID = (dx < 2 && dy >= 1) ? ((C::WM - 2 > 0 ? (C::WM - 2 + (dy == 1 ? 1 : 0)) / 2 : 0)) : 0;      // middle rows fetched under this tap
        constexpr int nloads = (dy == 0 ? 2 : 0) + ((dx < 2 || dy < 2) ? C::WN : 0) + ((dx < 2 && dy >= 1) ? 1 : 0) + NMID;
        if constexpr (dy == 0) {
#pragma unroll
          for (int r = C::WM; r < NP; ++r) load_p(dx, r);                              // rows first needed at dy >= 1
        }
        if constexpr (dx < 2 || dy < 2) load_w(dy == 2 ? dx + 1 : dx, dy == 2 ? 0 : dy + 1, wf[cur ^ 1]);
        if constexpr (dx < 2 && dy >= 1) {
          load_p(dx + 1, dy - 1);                                                      // row 0 under dy = 1, row 1 under dy = 2 (dead in this column by then)
#pragma unroll
          for (int r = 2; r <= C::WM - 1; ++r) if (((r - 2) & 1) == (dy - 1)) load_p(dx + 1, r);      // the double-buffered middle rows, spread over both taps
        }
#pragma unroll
        for (int n = 0; n < C::WN; ++n)
#pragma unroll
          for (int m = 0; m < C::WM; ++m) mma_step<EK>(acc[n][m], wf[cur][n], prow(dx, m + dy));
        // one ds_read behind each of the first MFMAs of the tap, the rest of the MFMAs after them (as in the generic form below)
        constexpr int NMT = C::WM * C::WN;
        constexpr int nl = nloads < NMT ? nloads : NMT;
        if constexpr (nl > 0) {
#pragma unroll
          for (int i = 0; i < nl - 1; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, nloads - nl + 1, 0);
        }
        if constexpr (NMT > nl) __builtin_amdgcn_sched_group_barrier(0x008, NMT - nl, 0);
      };
      using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
      tap(I0{}, I0{}); tap(I0{}, I1{}); tap(I0{}, I2{});
      tap(I1{}, I0{}); tap(I1{}, I1{}); tap(I1{}, I2{});
      tap(I2{}, I0{}); tap(I2{}, I1{}); tap(I2{}, I2{});
    } else
    {
'''
assert old in s
s=s.replace(old,new)
open(p,'w').write(s)
p='/root/repo/diffusiondepth_amd/csrc/dd_igemm2_cfg.h'
s=open(p).read()
old='''  static constexpr int C3 = (C3SHAPE && ESZ == 2 && !SPLIT) ? (BIG ? 2 : DD_C3) : 0;
'''
new=old+'''  // row-reuse form of the MFMA block (dd_igemm2.hip, mfma_block): the conv3-shaped layers in their nine-taps-per-stage tilings
  static constexpr bool DYR = DD_DY_REUSE && C3 != 0 && !PRED5 && ((C3 == 2) || DD_C3 == 1);
'''
assert old in s; s=s.replace(old,new)
old='''#ifndef DD_C3
#define DD_C3 1
#endif
'''
new=old+'''// conv3-shaped layers (nine taps of a 16-channel chunk per stage): 1 = the MFMA block walks the taps column by column and reuses each patch-row fragment
// for the three vertical taps (30 instead of 36 LDS fragment reads per 36 MFMAs on 8x32 tiles, 36 instead of 54 per 72 on 16x32 tiles); 0 = one
// (pixel, weight) fragment group per tap, row-major
#ifndef DD_DY_REUSE
#define DD_DY_REUSE 0
#endif
'''
assert old in s; s=s.replace(old,new)
open(p,'w').write(s)
print("patched")
