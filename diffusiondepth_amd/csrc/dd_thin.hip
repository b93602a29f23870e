// dd_thin.hip -- conv4 of the denoiser (64 -> 16, `model.pred.3`; reference src/model/head/ddim_depth_estimate_res.py:319-321) as a PERSISTENT
// STREAMING kernel.  (conv1 -- 16 -> 64 with the fused DDIM update -- was built the same way in round 3 and measured: 40.3 -> 38.8 us at
// B=4, 70.9 -> 66.2 at B=8, 15.8 -> 17.3 at B=1; not kept: profiles/history/r03_run2_variants.md.)
//
// conv4 carries 3 % of the step's FLOPs and is bound by memory: 192 algorithmic bytes per latent pixel (y3 in, y4 out) against 18 kFLOP.  As an
// instance of the general convolution kernel (dd_igemm2.hip, one workgroup per tile) it ran at 0.27-0.29 of the HBM rate: a workgroup's life
// is a serial chain -- launch, wait for the producer's GroupNorm sums, fp64 table, first patch, four chunk stages with a weight DMA each,
// stores, statistics atomics -- around 36 MFMAs, and every tile re-streams the 37 KB of weights (as much as its input patch) through LDS.
// Here:
//   * B x n workgroups (n per image, 512 in all = two per CU), each walking the tiles j, j + n, ... of ONE image: the GroupNorm table is
//     built once per workgroup, the 36 KB of packed weights are copied into LDS once and stay there, the statistics of all its tiles leave
//     in one set of atomics;
//   * rolling prefetch: the raw input of a whole tile (four 16-channel chunks) lives in registers; a chunk's registers are reloaded with the
//     NEXT tile's chunk as soon as the chunk has been normalised into LDS, so every load has a full tile period to land and ~43 KB per
//     workgroup are in flight all the time (Little: two workgroups per CU x 256 CUs x 43 KB = 22 MB against 8 TB/s x ~2.5 us);
//   * no counted waits: the input loads are ordinary loads tracked by the compiler (only the one-time weight copy is an LDS-DMA, waited for with vmcnt(0)); workgroup barriers are raw `s_barrier`s behind an
//     explicit `s_waitcnt lgkmcnt(0)` (a __syncthreads() would drain the prefetched loads: vmcnt(0)).
// Round 4, the refined-f16 mode (EK_F16R, dd_kernels.h) -- three template switches on the same pipeline:
//   STACK   the packed weight image carries an f16 PAIR per weight in ONE MFMA: the 32x32x16 instruction has 32 cout rows, conv4 has 16 couts, so rows
//           16..31 (zero padding in the plain image) hold f16((w - f16(w)) * 2^11) and the epilogue adds accumulator quads q and q + 2 (lane-local).
//           Same MFMA count, same LDS traffic; conv4's weight rounding -- the largest single term of the f16 mode's depth error -- is gone.
//   INQ     y3 arrives as int16 with one fp32 scale per pixel (EK_F16R, dd_kernels.h: f16's bytes, ~15 bits relative to the pixel's largest channel)
//           instead of the 2-byte kind: one more 4-byte load per staging item.  (An fp32 y3 -- two 16-byte loads per item, a two-slot register ring
//           to stay inside 128 VGPRs -- was built first and measured: same depth error, conv4 30 -> 45 us; removed: profiles/history/r04_call1_*.)
//   PSPLIT  the operand relu(gn3(y3)) as an f16 pair as well (scaled by 2^4 so that lo halves stay normal): a second LDS plane per patch buffer and
//           a second MFMA per tap against the same stacked weight fragment -- [Whi; Wlo] . (Phi + Plo) = all four partial products
// Same arithmetic as layer 4 of dd_igemm2.hip: the packed weight image of that layer (16-channel chunks, nine taps per stage, 32 cout rows,
// swizzle pre-applied) is used as it is, accumulation order per output is (chunk, tap), the GroupNorm table is computed by the same fp64
// expressions.  Only the order of the fp32 per-lane partial sums of the NEXT GroupNorm's statistics differs (more pixels per lane).
#include "dd_igemm2_cfg.h"
#include "dd_gcn.h"

namespace dd {

namespace thin {
constexpr int TH = 8, TW = 32, PH = TH + 2, PW = TW + 2;
constexpr int CIN = HID_C, COUT = LATENT_C, NROW = 32;          // 32 cout rows in the packed image (16 real)
constexpr int CK = 16, NCH = CIN / CK, ROWB = CK * 2, PPP = ROWB / 16, RPB = 256 / ROWB, EPP = 8;
constexpr int THREADS = 512, WAVES = 8;
constexpr int W_STAGE = 9 * NROW * ROWB;                        // 9216 B per chunk
constexpr int W_BYTES = NCH * W_STAGE;                          // 36864 B, resident
constexpr int PATCH_PLANE = PH * PW * ROWB;                     // 10880 B per chunk and operand plane
constexpr int ITEMS = PH * PW * PPP;                            // 680 sixteen-byte pieces per chunk
constexpr int NIT = (ITEMS + THREADS - 1) / THREADS;            // 2
constexpr int TAB_BYTES = (2 * CIN + NROW) * 4 + WAVES * 8 * 8;
constexpr int smem_bytes(bool psplit) { return W_BYTES + 2 * (psplit ? 2 : 1) * PATCH_PLANE + TAB_BYTES; }
static_assert(smem_bytes(true) <= 80 * 1024, "two workgroups per CU");
static_assert(NCH == 4 && NIT == 2 && (NCH % 2) == 0, "rolling prefetch below is written for four chunks");
static_assert(Cfg2<EK_F16, 4>::CK == CK && Cfg2<EK_F16, 4>::TG == 9 && Cfg2<EK_F16, 4>::NT == NROW && Cfg2<EK_F16, 4>::W_BYTES == W_STAGE,
              "the packed weight image of layer 4 (dd_igemm2_cfg.h, DD_C4_CK16) is read as it is");
}  // namespace thin

template <int EK, bool STACK, bool INQ, bool PSPLIT>
__global__ void __launch_bounds__(thin::THREADS, 4) conv4_stream_kernel(ConvParams p) {
  using namespace thin;
  static_assert(EK == EK_F16 || EK == EK_BF16, "2-byte kinds");
  static_assert(EK == EK_F16 || !(STACK || INQ || PSPLIT), "the refined forms are f16 kernels");
  static_assert(STACK || !PSPLIT, "the operand pair runs against the stacked weight image");
  constexpr int PATCH_BYTES = (PSPLIT ? 2 : 1) * PATCH_PLANE;
  constexpr int TAB_OFF = W_BYTES + 2 * PATCH_BYTES;
  constexpr float PSC = PSPLIT ? SPLIT_PSCALE : 1.f;   // the patch is carried times 2^4 (exact; relu commutes), the epilogue divides
  DD_DYN_SMEM(smem);
  float* tab_a = reinterpret_cast<float*>(smem + TAB_OFF);
  float* tab_b = tab_a + CIN;
  float* tab_bias = tab_b + CIN;
  double* s_red = reinterpret_cast<double*>(tab_bias + NROW);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, g = lane >> 5;
  const int h = p.h, w = p.w;
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  // timing experiments (-DDD_ABLATE=1 builds only; results are wrong): bit0 no patch transform / LDS writes, bit1 no prefetch loads, bit3 no fragment
  // reads / MFMAs, bit4 no output stores, bit5 no statistics atomics, bit6 no per-stage barrier, bit7 no weight copy
  const int abl = DD_ABLATE ? p.ablate : 0;
  // workgroup -> (image, first tile, stride): persist_grid() launches B * n workgroups.  (Round 5 measured an XCD-aware map here -- every XCD one
  // contiguous 4 x 16-tile block of an image, so that halo rows come out of its L2 instead of HBM twice: conv4 32.3 -> 34.3 us at KITTI B=4, i.e.
  // SLOWER, A/B/A/B on one box; the interleaved map spreads every moment's loads over all HBM channels and that matters more than the 20 MB of
  // halo re-reads: profiles/r05_experiments.md section 1.  Not kept.)
  const int wg = blockIdx.x, n_per_img = (int)gridDim.x / p.B;
  const int b = wg % p.B;
  int tl = wg / p.B;

  // ---- once per workgroup: weights -> LDS (the packed image IS the LDS image), bias, GroupNorm table of image b ----
  // Round 4: every load of the workgroup's start-up is ISSUED before anything waits -- the GroupNorm partial sums and gamma / beta (the fp64 table is
  // the longest dependent chain), the 36 KB of weights (LDS-DMA: no registers, no wait until the first barrier), and the first tile's raw input --
  // instead of three memory round trips one after the other (weights copied through registers, then the sums fetched, then the tile requested):
  // a workgroup walks only one to four tiles, so its start-up is 20 .. 40 % of its life.
  const double* st = p.stats_in + (size_t)b * STAT_SLOTS * STAT_STRIDE + (lane >> 1) * STAT_STRIDE + (lane & 1) * 4;
  double2 sv0 = *reinterpret_cast<const double2*>(st), sv1 = *reinterpret_cast<const double2*>(st + 2);
  const int ct = tid < CIN ? tid : 0;                   // (unconditional, clamped: no branch ties a wait to these loads)
  const float my_gamma = p.gn_gamma[ct], my_beta = p.gn_beta[ct];
  const float my_bias = p.bias[tid < NROW ? tid : 0];
  if (!(abl & 128)) {
    static_assert(W_BYTES % 1024 == 0, "whole 1-KiB DMA pieces");
    const unsigned lds_base = DD_LDS_BASE(smem);
    for (int kc = wave; kc < W_BYTES / 1024; kc += WAVES) {
      const char* gsrc = reinterpret_cast<const char*>(p.wpack) + (size_t)kc * 1024 + lane * 16;
      const unsigned ldst = __builtin_amdgcn_readfirstlane(lds_base + kc * 1024);
      DD_LDS_DMA16(smem, gsrc, ldst);
    }
  }
  auto finish_startup = [&]() {
    if (tid < NROW) tab_bias[tid] = my_bias;
#pragma unroll
    for (int off = 2; off <= 32; off <<= 1) {
      sv0.x += __shfl_xor(sv0.x, off, 64); sv0.y += __shfl_xor(sv0.y, off, 64);
      sv1.x += __shfl_xor(sv1.x, off, 64); sv1.y += __shfl_xor(sv1.y, off, 64);
    }
    const double2 ov0 = make_double2(__shfl_xor(sv0.x, 1, 64), __shfl_xor(sv0.y, 1, 64));
    const double2 ov1 = make_double2(__shfl_xor(sv1.x, 1, 64), __shfl_xor(sv1.y, 1, 64));
    const bool hi = lane & 1;
    const double2 g0 = hi ? ov0 : sv0, g1 = hi ? ov1 : sv1, g2 = hi ? sv0 : ov0, g3 = hi ? sv1 : ov1;
    // non-finite statistics of y3: the image is poisoned (dd_igemm2.hip, same rule) -- conv4's bias becomes NaN, y4 and its statistics with it
    const bool poisoned = !(((g0.x - g0.x) + (g0.y - g0.y)) + ((g1.x - g1.x) + (g1.y - g1.y)) + ((g2.x - g2.x) + (g2.y - g2.y)) + ((g3.x - g3.x) + (g3.y - g3.y)) == 0.0);
    if (poisoned && tid < NROW) tab_bias[tid] = __builtin_nanf("");
    if (tid < CIN) {
      constexpr int CG = CIN / GN_GROUPS;
      const int grp = tid / CG;
      const double2 gs = grp == 0 ? g0 : grp == 1 ? g1 : grp == 2 ? g2 : g3;
      const double inv_cnt = 1.0 / ((double)h * (double)w * (double)CG);
      const double mean = gs.x * inv_cnt;
      double var = gs.y * inv_cnt - mean * mean;
      var = var > 0.0 ? var : 0.0;
      const double a = (double)my_gamma / sqrt(var + (double)GN_EPS);
      tab_a[tid] = (float)a * PSC;
      tab_b[tid] = (float)((double)my_beta - mean * a) * PSC;
    }
  };

  // ---- per-thread staging items (independent of tile and chunk): piece jfix of patch pixel pp(u) ----
  const int jfix = tid & (PPP - 1);
  int lds_off[NIT], pr_[NIT], pc_[NIT];
  unsigned m_valid = 0;
#pragma unroll
  for (int u = 0; u < NIT; ++u) {
    const int it = u * THREADS + tid;
    const int itc = it < ITEMS ? it : ITEMS - 1;
    const int pp = itc >> 1;
    pr_[u] = pp / PW; pc_[u] = pp - pr_[u] * PW;
    if (it < ITEMS) m_valid |= 1u << u;
    lds_off[u] = pp * ROWB + ((jfix << 4) ^ swz16<RPB, PPP>(pc_[u]));
  }
  const char* in_b = reinterpret_cast<const char*>(p.in) + (size_t)b * h * w * CIN * 2;
  // geometry of a tile: clamped pixel offsets of this thread's items and their inside-the-image mask
  auto geometry = [&](int t_local, int* po, unsigned& mi) {
    const int ty = t_local / p.tiles_x, tx = t_local - ty * p.tiles_x;
    mi = 0;
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      const int gy = ty * TH - 1 + pr_[u], gx = tx * TW - 1 + pc_[u];
      if (gy >= 0 && gy < h && gx >= 0 && gx < w) mi |= 1u << u;
      const int gyc = gy < 0 ? 0 : (gy >= h ? h - 1 : gy), gxc = gx < 0 ? 0 : (gx >= w ? w - 1 : gx);
      po[u] = gyc * w + gxc;
    }
  };
  int po_cur[NIT], po_nx[NIT];
  unsigned mi_cur = 0, mi_nx = 0;
  geometry(tl, po_cur, mi_cur);
  // The prefetch loads of the "next tile" are issued UNCONDITIONALLY (hipcc counts its vmcnt waits only through straight-line load / use
  // sequences: loads behind `if (has_next)` made it fall back to vmcnt(0/1) in every stage).  Behind the last tile they all read pixel 0 of
  // the image (one cache line per wave, discarded).
  auto next_geometry = [&]() {
    if (tl + n_per_img < tiles_per_img) { geometry(tl + n_per_img, po_nx, mi_nx); return true; }
#pragma unroll
    for (int u = 0; u < NIT; ++u) po_nx[u] = 0;
    mi_nx = 0;
    return false;
  };
  bool has_next = next_geometry();

  // raw-input registers: one slot per chunk, a chunk's slot is refilled with the NEXT tile's same chunk (four stages ahead)
  uint4 raw[NCH][NIT];
  float rsc[INQ ? NCH : 1][NIT];                                 // INQ: the pixel's scale (travels with the item)
  const float* sc_b = INQ ? p.cadd_scale + (size_t)b * h * w : nullptr;
  auto load_chunk = [&](int c, const int* po) {        // channel-blocked y3: [B][2][h][w][32]
    const int cbase = c * CK + jfix * EPP;
    const size_t off0 = (size_t)(cbase >> 5) * h * w * ACT_CB + (cbase & (ACT_CB - 1));
    if (abl & 2) return;
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      raw[c][u] = *reinterpret_cast<const uint4*>(in_b + (off0 + (size_t)po[u] * ACT_CB) * 2);
      if constexpr (INQ) rsc[c][u] = sc_b[po[u]];
    }
  };
  auto transform_chunk = [&](int c, int buf, unsigned mi) {        // relu(gn3(y3)) of chunk c -> patch buffer `buf`
    if (abl & 1) return;
    float ta[EPP], tb[EPP];
    const int c0 = c * CK + jfix * EPP;
#pragma unroll
    for (int q = 0; q < EPP / 4; ++q) {
      const float4 a4 = *reinterpret_cast<const float4*>(tab_a + c0 + 4 * q), b4 = *reinterpret_cast<const float4*>(tab_b + c0 + 4 * q);
      ta[4 * q] = a4.x; ta[4 * q + 1] = a4.y; ta[4 * q + 2] = a4.z; ta[4 * q + 3] = a4.w;
      tb[4 * q] = b4.x; tb[4 * q + 1] = b4.y; tb[4 * q + 2] = b4.z; tb[4 * q + 3] = b4.w;
    }
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      // The raw registers are consumed UNCONDITIONALLY (selects, no branch around the use): a use inside a branch leaves the load "pending" on
      // the other path in hipcc's vmcnt bookkeeping, and the unconditional reload of the same registers below then waits vmcnt(0/1) in every
      // stage -- which drains the whole prefetch (seen in the first version of this kernel: conv4 30 us instead of 36, not 22).
      const bool inside = (mi >> u) & 1u;                   // zero padding applies AFTER the normalisation
      uint4 v, vlo = make_uint4(0u, 0u, 0u, 0u);
      if constexpr (!INQ && !PSPLIT) {
        v = affine_relu_pack<EK, EK>(raw[c][u], ta, tb);
      } else {
        float y[EPP];
        if constexpr (INQ) {
          const uint32_t qw[4] = {raw[c][u].x, raw[c][u].y, raw[c][u].z, raw[c][u].w};
          const float sc = rsc[c][u];
#pragma unroll
          for (int i = 0; i < 4; ++i) { y[2 * i] = (float)(short)(qw[i] & 0xFFFFu) * sc; y[2 * i + 1] = (float)((int)qw[i] >> 16) * sc; }
        } else {
          Piece<EK>::unpack(raw[c][u], y);
        }
#pragma unroll
        for (int i = 0; i < EPP; ++i) y[i] = fmaxf(fmaf(ta[i], y[i], tb[i]), 0.f);
        v = Piece<EK>::pack(y);
        if constexpr (PSPLIT) {      // operand pair: hi = f16(a), lo = f16(a - hi)
          float yh[EPP];
          Piece<EK>::unpack(v, yh);
#pragma unroll
          for (int i = 0; i < EPP; ++i) yh[i] = y[i] - yh[i];
          vlo = Piece<EK>::pack(yh);
          vlo.x = inside ? vlo.x : 0u; vlo.y = inside ? vlo.y : 0u; vlo.z = inside ? vlo.z : 0u; vlo.w = inside ? vlo.w : 0u;
        }
      }
      v.x = inside ? v.x : 0u; v.y = inside ? v.y : 0u; v.z = inside ? v.z : 0u; v.w = inside ? v.w : 0u;
      if ((m_valid >> u) & 1u) {
        *reinterpret_cast<uint4*>(smem + W_BYTES + buf * PATCH_BYTES + lds_off[u]) = v;
        if constexpr (PSPLIT) *reinterpret_cast<uint4*>(smem + W_BYTES + buf * PATCH_BYTES + PATCH_PLANE + lds_off[u]) = vlo;
      }
    }
  };

  // first tile: all four chunks requested (behind the start-up loads, in front of their first use), then the table; the weight DMA (invisible to
  // hipcc's vmcnt bookkeeping) has landed when nothing is outstanding any more -- the first tile's input is needed right behind the barrier anyway
#pragma unroll
  for (int c = 0; c < NCH; ++c) load_chunk(c, po_cur);
  finish_startup();
  DD_WAIT_VM(0);
  DD_WAIT_LGKM0();
  __builtin_amdgcn_s_barrier();                          // table + weights + bias visible
  asm volatile("" ::: "memory");
  transform_chunk(0, 0, mi_cur);
  DD_WAIT_LGKM0();
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // per-lane fragment addresses: pixel block = tile row `wave`, lane li = tile column
  int colt[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) colt[dx] = W_BYTES + (wave * PW + li + dx) * ROWB + ((g << 4) ^ swz16<RPB, PPP>(li + dx));
  const int wkt = li * ROWB + ((g << 4) ^ swz16<RPB, PPP>(li));

  float ls[2] = {0.f, 0.f}, lq[2] = {0.f, 0.f};          // GroupNorm-4 partial sums of this lane: groups {g, 2 + g}
  char* out_b = reinterpret_cast<char*>(p.out) + (size_t)b * h * w * COUT * 4;
#pragma unroll 1
  for (;;) {
    f32x16_t acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // the slot of chunk 0 was consumed in the previous tile's last stage: refill it with the next tile's chunk 0
      if (c == 0) load_chunk(0, po_nx);
      const int pbuf = (c & 1) * PATCH_BYTES;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if (abl & 8) break;
        const int dy = t / 3, dx = t % 3;
        const uint4 pf = *reinterpret_cast<const uint4*>(smem + colt[dx] + pbuf + dy * (PW * ROWB));
        const uint4 wf = *reinterpret_cast<const uint4*>(smem + wkt + c * W_STAGE + t * (NROW * ROWB));
        mma_step<EK>(acc, wf, pf);
        if constexpr (PSPLIT) {
          const uint4 pl = *reinterpret_cast<const uint4*>(smem + colt[dx] + pbuf + PATCH_PLANE + dy * (PW * ROWB));
          mma_step<EK>(acc, wf, pl);
        }
      }
      // stage the following chunk into the other patch buffer, then reuse its registers for the next tile
      if (c + 1 < NCH) {
        transform_chunk(c + 1, (c + 1) & 1, mi_cur);
        load_chunk(c + 1, po_nx);
      } else {
        transform_chunk(0, 0, mi_nx);          // (behind the last tile: a patch nobody reads)
      }
      DD_WAIT_LGKM0();
      if (!(abl & 64)) __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    // ---- epilogue of this tile: bias, statistics, fp32 NHWC store (lane: pixel (wave, li), couts 4g..4g+3 and 8+4g..8+4g+3) ----
    {
      const int ty = tl / p.tiles_x, tx = tl - ty * p.tiles_x;
      const int gy = ty * TH + wave, gx = tx * TW + li;
      if (gy < h && gx < w) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float4 bv = *reinterpret_cast<const float4*>(tab_bias + 8 * q + 4 * g);
          float v0, v1, v2, v3;
          if constexpr (STACK) {
            // cout c = 8q + 4g + i sits in quad q, its lo-half row 16 + c in quad q + 2 of the SAME lane: (hi + lo * 2^-11) / patch scale + bias
            constexpr float LS = 1.f / STACK_LSCALE, OS = 1.f / PSC;
            v0 = fmaf(fmaf(acc[(q + 2) * 4 + 0], LS, acc[q * 4 + 0]), OS, bv.x); v1 = fmaf(fmaf(acc[(q + 2) * 4 + 1], LS, acc[q * 4 + 1]), OS, bv.y);
            v2 = fmaf(fmaf(acc[(q + 2) * 4 + 2], LS, acc[q * 4 + 2]), OS, bv.z); v3 = fmaf(fmaf(acc[(q + 2) * 4 + 3], LS, acc[q * 4 + 3]), OS, bv.w);
          } else {
            v0 = acc[q * 4 + 0] + bv.x; v1 = acc[q * 4 + 1] + bv.y; v2 = acc[q * 4 + 2] + bv.z; v3 = acc[q * 4 + 3] + bv.w;
          }
          ls[q] += (v0 + v1) + (v2 + v3);
          lq[q] += fmaf(v0, v0, fmaf(v1, v1, fmaf(v2, v2, v3 * v3)));
          // (hidden from hipcc's vmcnt bookkeeping: a tracked store beside the prefetched loads would turn every later wait into vmcnt(0))
          if (!(abl & 16)) DD_GLOBAL_STORE16_UNTRACKED(out_b + (((size_t)gy * w + gx) * COUT + 8 * q + 4 * g) * 4, make_float4(v0, v1, v2, v3));
        }
      }
    }
    if (!has_next) break;
    tl += n_per_img;
#pragma unroll
    for (int u = 0; u < NIT; ++u) po_cur[u] = po_nx[u];
    mi_cur = mi_nx;
    has_next = next_geometry();
  }
  // ---- GroupNorm-4 statistics of all tiles of this workgroup: butterfly inside each half-wave, fp64 across waves, one atomic per value ----
#pragma unroll
  for (int off = 16; off >= 1; off >>= 1) {
#pragma unroll
    for (int k = 0; k < 2; ++k) { ls[k] += __shfl_xor(ls[k], off, 64); lq[k] += __shfl_xor(lq[k], off, 64); }
  }
  if (li == 0) {
    s_red[wave * 8 + (0 + g) * 2 + 0] = (double)ls[0]; s_red[wave * 8 + (0 + g) * 2 + 1] = (double)lq[0];
    s_red[wave * 8 + (2 + g) * 2 + 0] = (double)ls[1]; s_red[wave * 8 + (2 + g) * 2 + 1] = (double)lq[1];
  }
  __syncthreads();
  if (tid < 8 && !(abl & 32)) {
    double tot = 0.0;
#pragma unroll
    for (int wv = 0; wv < WAVES; ++wv) tot += s_red[wv * 8 + tid];
    atomicAdd(p.stats_out + ((size_t)b * STAT_SLOTS + (wg % STAT_SLOTS)) * STAT_STRIDE + tid, tot);
  }
}

template <int EK, bool STACK, bool INQ, bool PSPLIT> static hipError_t launch_conv4_stream_k(const ConvParams& p, hipStream_t s) {
  static bool attr_set = false;
  constexpr int SMEM = thin::smem_bytes(PSPLIT);
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv4_stream_kernel<EK, STACK, INQ, PSPLIT>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const unsigned n_wg = (unsigned)persist_grid(p.B, p.tiles_x * p.tiles_y, p.persist_slots > 0 ? p.persist_slots : 512);
  hipLaunchKernelGGL((conv4_stream_kernel<EK, STACK, INQ, PSPLIT>), dim3(n_wg), dim3(thin::THREADS), SMEM, s, p);
  return hipGetLastError();
}

hipError_t launch_conv4_stream(int ek, const ConvParams& p, hipStream_t s, bool stack, bool in_q15, bool psplit) {
  if (ek == EK_F16 && stack) {
    if (in_q15) return psplit ? launch_conv4_stream_k<EK_F16, true, true, true>(p, s) : launch_conv4_stream_k<EK_F16, true, true, false>(p, s);
    return psplit ? launch_conv4_stream_k<EK_F16, true, false, true>(p, s) : launch_conv4_stream_k<EK_F16, true, false, false>(p, s);
  }
  if (stack || in_q15 || psplit) return hipErrorInvalidValue;
  switch (ek) {
    case EK_F16: return launch_conv4_stream_k<EK_F16, false, false, false>(p, s);
    case EK_BF16: return launch_conv4_stream_k<EK_BF16, false, false, false>(p, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace dd
