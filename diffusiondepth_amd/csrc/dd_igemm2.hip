// dd_igemm2.hip -- v2 of the fused 3x3 convolution (see dd_igemm.hip for the algorithm): same tiling
// and MFMA mapping, re-structured as a software pipeline.
//
//   * weights never touch VGPRs: each (channel-chunk, tap-group) stage is copied global -> LDS with
//     `global_load_lds_dwordx4` (LDS-DMA, 1 KiB per wave-instruction) into a 2-deep ring, issued one
//     stage ahead of the MFMAs that consume it.  The DMA writes LDS linearly, so the bank-conflict
//     swizzle is applied to the packed GLOBAL image on the host (cdna guide rule 21) and undone by
//     the same XOR on the ds_read side.
//   * the halo'd input patch of the NEXT channel chunk is fetched into registers at the start of a
//     chunk, and normalised (GroupNorm affine + ReLU [+ cond + E[t]]) / written to the other patch
//     buffer behind the last tap group of the current chunk: HBM latency and the prologue VALU work
//     hide under MFMAs.  Loads are unconditional (clamped coordinates) so every wave issues the same
//     number of VMEM instructions.
//   * LDS rows are unpadded (32 / 64 / 128 B) with a 16-B-piece XOR swizzle
//         piece' = piece ^ ((row / R) & (PPP-1)),  R = 256 / ROWB,  PPP = ROWB / 16
//     which makes every 16-lane ds_read_b128 group hit 16 distinct 16-B slots of the 256-B bank row
//     for ANY run of 32 consecutive rows (rows mod 16 <-> slots is a bijection), and keeps
//     conv2 / conv3 at <= 77 KiB of LDS (two workgroups per CU).
//   * bf16 / f16 outputs are stored 16 B per lane: `v_permlane32_swap` pairs the two half-waves'
//     4-cout quads into 8 consecutive couts (cdna guide T21).
#include "dd_elem.h"

namespace dd {

template <int EK_, int LAYER_> struct Cfg2 {
  static constexpr int EK = EK_;
  static constexpr int LAYER = LAYER_;
  static constexpr int ESZ = ElemSize<EK>::V;
  static constexpr int CIN = (LAYER == 1) ? LATENT_C : (LAYER == 2) ? HID_C : (LAYER == 3) ? COND_C : HID_C;
  static constexpr int COUT = (LAYER == 1) ? HID_C : (LAYER == 2) ? COND_C : (LAYER == 3) ? HID_C : LATENT_C;
  static constexpr int COUT_PAD = (COUT < 32) ? 32 : COUT;
  static constexpr int CK = (LAYER == 1) ? 16 : (LAYER == 3) ? (64 / ESZ) : (128 / ESZ);
  static constexpr int TG = (LAYER == 1) ? 9 : (LAYER == 2) ? 1 : 3;
  static constexpr int NT = (LAYER == 2) ? 128 : COUT_PAD;
  static constexpr int TH = 8, TW = 32;
  static constexpr int WAVES = 4;
  static constexpr int THREADS = WAVES * 64;
  static constexpr int WM = (TH * TW) / (32 * WAVES);
  static constexpr int WN = NT / 32;
  static constexpr int PRO = (LAYER == 1) ? PRO_X : (LAYER == 3) ? PRO_GN_ADD : PRO_GN;
  static constexpr int IN_ESZ = (LAYER == 1) ? 4 : ESZ;
  static constexpr int OUT_ESZ = (LAYER == 4) ? 4 : ESZ;
  static constexpr int PH = TH + 2, PW = TW + 2;
  static constexpr int ROWB = CK * ESZ;                  // 32 / 64 / 128 bytes
  static constexpr int PPP = ROWB / 16;
  static constexpr int RPB = 256 / ROWB;                 // rows per 256-B LDS bank row
  static constexpr int EPP = 16 / ESZ;
  static constexpr int NCHUNK = CIN / CK;
  static constexpr int NTG = 9 / TG;
  static constexpr int NSTAGE = NCHUNK * NTG;
  static constexpr int NPB = (NCHUNK > 1) ? 2 : 1;       // patch buffers
  static constexpr int PATCH_BYTES = PH * PW * ROWB;
  static constexpr int W_BYTES = TG * NT * ROWB;
  static constexpr int CTAB = (LAYER == 1) ? LATENT_C : CIN;
  static constexpr int TAB_FLOATS = 3 * CTAB + 8;
  static constexpr int SMEM_BYTES = NPB * PATCH_BYTES + 2 * W_BYTES + TAB_FLOATS * 4;
  static constexpr int ITEMS = PH * PW * PPP;
  static constexpr int NIT = (ITEMS + THREADS - 1) / THREADS;      // staging items per thread
  static constexpr int NLD = EPP * IN_ESZ / 16;
  // two workgroups per CU (8 waves = 2 per SIMD) need <= 256 VGPR+AGPR; the fp32 layer-1 tile needs 96 KiB of LDS -> 1 per CU
  static constexpr int MIN_WAVES_PER_SIMD = (SMEM_BYTES <= 80 * 1024) ? 2 : 1;
  static_assert(CIN % CK == 0 && 9 % TG == 0 && COUT_PAD % NT == 0, "tiling");
  static_assert(PATCH_BYTES % 16 == 0 && W_BYTES % 1024 == 0, "LDS carve / DMA granularity");
  static_assert(PATCH_BYTES >= STAT_SLOTS * 8 * 8 + 64 + WAVES * 8 * 8, "scratch fits in the patch region");
  static_assert(ROWB == 32 || ROWB == 64 || ROWB == 128, "swizzle derivation");
};

__device__ __forceinline__ int swz_of_row(int row, int rpb, int ppp) { return (row / rpb) & (ppp - 1); }

template <class C>
__global__ void __launch_bounds__(C::THREADS, C::MIN_WAVES_PER_SIMD) conv_igemm2_kernel(ConvParams p) {
  constexpr int EK = C::EK;
  constexpr int PW = C::PW, ROWB = C::ROWB, PPP = C::PPP, RPB = C::RPB, EPP = C::EPP, CK = C::CK;
  constexpr int NIT = C::NIT, NLD = C::NLD, IN_ESZ = C::IN_ESZ;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* s_patch = smem;                                           // NPB buffers
  char* s_w = smem + C::NPB * C::PATCH_BYTES;                     // 2 buffers
  float* s_tab = reinterpret_cast<float*>(smem + C::NPB * C::PATCH_BYTES + 2 * C::W_BYTES);
  float* tab_a = s_tab;
  float* tab_b = s_tab + C::CTAB;
  float* tab_e = s_tab + 2 * C::CTAB;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, g = lane >> 5;

  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const int b = blockIdx.x / tiles_per_img;
  const int trem = blockIdx.x - b * tiles_per_img;
  const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
  const int y0 = ty * C::TH, x0 = tx * C::TW;
  const int n0 = blockIdx.y * C::NT;
  const int h = p.h, w = p.w;
  const bool have_norm = (C::PRO != PRO_X) || (p.step > 0);

  const char* in_b = reinterpret_cast<const char*>(p.in) + (size_t)b * h * w * C::CIN * IN_ESZ;
  const char* cond_b = (C::PRO == PRO_GN_ADD) ? reinterpret_cast<const char*>(p.cond) + (size_t)b * h * w * C::CIN * IN_ESZ : nullptr;
  const char* y4_b = (C::PRO == PRO_X) ? reinterpret_cast<const char*>(p.y4) + (size_t)b * h * w * LATENT_C * 4 : nullptr;
  char* xout_b = (C::PRO == PRO_X) ? reinterpret_cast<char*>(p.xout) + (size_t)b * h * w * LATENT_C * 4 : nullptr;

  // ---- weight stage s -> ring slot (s & 1) by LDS-DMA: wave `wave` copies KiB-chunks wave, wave+4, ... ----
  auto issue_weights = [&](int s) {
    const char* src = reinterpret_cast<const char*>(p.wpack) + ((size_t)blockIdx.y * C::NSTAGE + s) * (size_t)C::W_BYTES;
    char* dst = s_w + (s & 1) * C::W_BYTES;
#pragma unroll
    for (int c = 0; c < (C::W_BYTES / 1024 + C::WAVES - 1) / C::WAVES; ++c) {
      const int kc = c * C::WAVES + wave;
      if (kc < C::W_BYTES / 1024) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)kc * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void*)(dst + kc * 1024), 16, 0, 0);
      }
    }
  };

  // ---- raw patch fetch of one channel chunk into registers (unconditional, clamped addresses) --------
  uint4 raw[NIT][NLD];
  uint4 aux[NIT][NLD];
  auto load_raw = [&](int chunk) {
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      int it = u * C::THREADS + tid;
      it = it < C::ITEMS ? it : C::ITEMS - 1;
      const int pp = it / PPP, j = it - pp * PPP;
      const int pr = pp / PW, pc = pp - pr * PW;
      int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
      gy = gy < 0 ? 0 : (gy >= h ? h - 1 : gy);
      gx = gx < 0 ? 0 : (gx >= w ? w - 1 : gx);
      const size_t goff = act_offset(C::CIN, h, w, 0, chunk * CK + j * EPP, gy, gx) * IN_ESZ;
#pragma unroll
      for (int q = 0; q < NLD; ++q) raw[u][q] = *reinterpret_cast<const uint4*>(in_b + goff + q * 16);
      if constexpr (C::PRO == PRO_GN_ADD) {
        aux[u][0] = *reinterpret_cast<const uint4*>(cond_b + goff);
      } else if constexpr (C::PRO == PRO_X) {
        if (have_norm) {
#pragma unroll
          for (int q = 0; q < NLD; ++q) aux[u][q] = *reinterpret_cast<const uint4*>(y4_b + goff + q * 16);
        }
      }
    }
  };

  float c1 = 1.f, c2 = 0.f;
  // ---- registers -> normalise -> swizzled LDS patch buffer ------------------------------------------
  auto transform_write = [&](int chunk, char* pbuf) {
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      const int it = u * C::THREADS + tid;
      if (it < C::ITEMS) {
        const int pp = it / PPP, j = it - pp * PPP;
        const int pr = pp / PW, pc = pp - pr * PW;
        const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
        const bool inside = gy >= 0 && gy < h && gx >= 0 && gx < w;
        float v[EPP];
        if (inside) {
          const int c0 = (C::PRO == PRO_X) ? j * EPP : chunk * CK + j * EPP;
          if constexpr (C::PRO == PRO_X) {
#pragma unroll
            for (int q = 0; q < NLD; ++q) {
              const uint32_t xw[4] = {raw[u][q].x, raw[u][q].y, raw[u][q].z, raw[u][q].w};
              const uint32_t yw[4] = {aux[u][q].x, aux[u][q].y, aux[u][q].z, aux[u][q].w};
#pragma unroll
              for (int i = 0; i < 4; ++i) {
                float x = __builtin_bit_cast(float, xw[i]);
                if (have_norm) {
                  const float e = fmaxf(fmaf(tab_a[c0 + q * 4 + i], __builtin_bit_cast(float, yw[i]), tab_b[c0 + q * 4 + i]), 0.f);
                  x = c1 * x + c2 * e;
                }
                v[q * 4 + i] = x;
              }
            }
            if (have_norm && pr >= 1 && pr <= C::TH && pc >= 1 && pc <= C::TW) {
              const size_t goff = act_offset(C::CIN, h, w, 0, j * EPP, gy, gx) * 4;
#pragma unroll
              for (int q = 0; q < NLD; ++q)
                *reinterpret_cast<float4*>(xout_b + goff + q * 16) = make_float4(v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]);
            }
          } else {
            Piece<EK>::unpack(raw[u][0], v);
#pragma unroll
            for (int i = 0; i < EPP; ++i) v[i] = fmaxf(fmaf(tab_a[c0 + i], v[i], tab_b[c0 + i]), 0.f);
            if constexpr (C::PRO == PRO_GN_ADD) {
              float cv[EPP];
              Piece<EK>::unpack(aux[u][0], cv);
#pragma unroll
              for (int i = 0; i < EPP; ++i) v[i] = v[i] + (cv[i] + tab_e[c0 + i]);
            }
          }
        } else {
#pragma unroll
          for (int i = 0; i < EPP; ++i) v[i] = 0.f;       // zero padding applies AFTER the normalisation
        }
        *reinterpret_cast<uint4*>(pbuf + pp * ROWB + ((j ^ swz_of_row(pp, RPB, PPP)) << 4)) = Piece<EK>::pack(v);
      }
    }
  };

  // ---- kick off: weights of stage 0 and the raw patch of chunk 0 -------------------------------------
  issue_weights(0);
  load_raw(0);

  // ---- GroupNorm affine table of the producing layer (overlaps the loads above) ----------------------
  if (have_norm) {
    double* s_tmp = reinterpret_cast<double*>(s_patch);
    const double* st = p.stats_in + (size_t)b * STAT_SLOTS * STAT_STRIDE;
    if (tid < STAT_SLOTS * 8) s_tmp[tid] = st[(tid >> 3) * STAT_STRIDE + (tid & 7)];
    __syncthreads();
    if (tid < 8) {
      double acc = 0.0;
      for (int s = 0; s < STAT_SLOTS; ++s) acc += s_tmp[s * 8 + tid];
      s_tmp[STAT_SLOTS * 8 + tid] = acc;
    }
    __syncthreads();
    constexpr int CG = C::CTAB / GN_GROUPS;
    const double cnt = (double)h * (double)w * (double)CG;
    for (int c = tid; c < C::CTAB; c += C::THREADS) {
      const int grp = c / CG;
      const double mean = s_tmp[STAT_SLOTS * 8 + grp * 2] / cnt;
      double var = s_tmp[STAT_SLOTS * 8 + grp * 2 + 1] / cnt - mean * mean;
      var = var > 0.0 ? var : 0.0;
      const double rstd = 1.0 / sqrt(var + (double)GN_EPS);
      const double a = (double)p.gn_gamma[c] * rstd;
      tab_a[c] = (float)a;
      tab_b[c] = (float)((double)p.gn_beta[c] - mean * a);
      if constexpr (C::PRO == PRO_GN_ADD) {
        const long long t = p.tvec[p.t_base + b * p.t_bstride];
        tab_e[c] = p.emb[(size_t)t * COND_C + c];
      }
    }
    if constexpr (C::PRO == PRO_X) { c1 = p.c1c2[2 * (p.step - 1)]; c2 = p.c1c2[2 * (p.step - 1) + 1]; }
    __syncthreads();                       // table visible; s_tmp (patch buffer 0) free again
  }
  transform_write(0, s_patch);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMA pieces of weight stage 0 have landed
  __syncthreads();                                    // patch 0 and weight stage 0 are in LDS for everybody

  f32x16_t acc[C::WN][C::WM];
#pragma unroll
  for (int n = 0; n < C::WN; ++n)
#pragma unroll
    for (int m = 0; m < C::WM; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.f;

  // per-lane addressing: patch row of (pixel block m, tap (0,0)); weight row offsets with the swizzle folded in
  int prow0[C::WM];
#pragma unroll
  for (int m = 0; m < C::WM; ++m) {
    const int pix = (wave * C::WM + m) * 32 + li;
    const int r = pix / C::TW, c = pix - r * C::TW;
    prow0[m] = r * PW + c;
  }
  const int g16 = g << 4;
  int wko[PPP / 2];                                   // weight piece offsets per k-step (row = .. + li: swizzle depends on li only)
  {
    const int wsw = swz_of_row(li, RPB, PPP) << 4;
#pragma unroll
    for (int kq = 0; kq < PPP / 2; ++kq) wko[kq] = li * ROWB + (((kq << 5) | g16) ^ wsw);
  }

  for (int chunk = 0; chunk < C::NCHUNK; ++chunk) {
    const char* pbuf = s_patch + (chunk & (C::NPB - 1)) * C::PATCH_BYTES;
#pragma unroll 1
    for (int tg = 0; tg < C::NTG; ++tg) {
      const int s = chunk * C::NTG + tg;
      if (s + 1 < C::NSTAGE) issue_weights(s + 1);
      if (C::NCHUNK > 1 && tg == 0 && chunk + 1 < C::NCHUNK) load_raw(chunk + 1);
      const char* wbuf = s_w + (s & 1) * C::W_BYTES;
#pragma unroll
      for (int t = 0; t < C::TG; ++t) {
        const int tap = tg * C::TG + t;
        const int dy = tap / 3, dx = tap - dy * 3;
        int pa[C::WM], psw[C::WM];
#pragma unroll
        for (int m = 0; m < C::WM; ++m) {
          const int row = prow0[m] + dy * PW + dx;
          pa[m] = row * ROWB;
          psw[m] = swz_of_row(row, RPB, PPP) << 4;
        }
#pragma unroll
        for (int kq = 0; kq < PPP / 2; ++kq) {
          uint4 pf[C::WM], wf[C::WN];
#pragma unroll
          for (int m = 0; m < C::WM; ++m)
            pf[m] = *reinterpret_cast<const uint4*>(pbuf + pa[m] + (((kq << 5) | g16) ^ psw[m]));
#pragma unroll
          for (int n = 0; n < C::WN; ++n)
            wf[n] = *reinterpret_cast<const uint4*>(wbuf + (t * C::NT + n * 32) * ROWB + wko[kq]);
#pragma unroll
          for (int n = 0; n < C::WN; ++n)
#pragma unroll
            for (int m = 0; m < C::WM; ++m) mma_step<EK>(acc[n][m], wf[n], pf[m]);
        }
      }
      if (C::NCHUNK > 1 && tg == C::NTG - 1 && chunk + 1 < C::NCHUNK)
        transform_write(chunk + 1, s_patch + ((chunk + 1) & (C::NPB - 1)) * C::PATCH_BYTES);
      // next stage's weights landed (this wave's DMA pieces), our patch writes retired; then everybody's
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // ---- epilogue: bias, GroupNorm partial sums, store --------------------------------------------------
  constexpr int NG_LOCAL = (C::COUT == COND_C) ? C::NT / (COND_C / GN_GROUPS) : 4;
  float ls[4] = {0.f, 0.f, 0.f, 0.f}, lq[4] = {0.f, 0.f, 0.f, 0.f};
  char* out_b = reinterpret_cast<char*>(p.out) + (size_t)b * h * w * C::COUT * C::OUT_ESZ;
#pragma unroll
  for (int m = 0; m < C::WM; ++m) {
    const int pix = (wave * C::WM + m) * 32 + li;
    const int r = pix / C::TW, c = pix - r * C::TW;
    const int gy = y0 + r, gx = x0 + c;
    const bool pvalid = gy < h && gx < w;
#pragma unroll
    for (int n = 0; n < C::WN; ++n) {
      uint2 pk[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (C::COUT < 32 && q >= 2) continue;            // conv4: couts 16..31 are zero padding
        const int co = n0 + n * 32 + 8 * q + 4 * g;
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + co);
        const float v[4] = {acc[n][m][q * 4 + 0] + bv.x, acc[n][m][q * 4 + 1] + bv.y,
                            acc[n][m][q * 4 + 2] + bv.z, acc[n][m][q * 4 + 3] + bv.w};
        if (pvalid) {
          const float s = (v[0] + v[1]) + (v[2] + v[3]);
          const float sq = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
          const int lg = (C::COUT == COND_C) ? (n >> 1) : (C::COUT == HID_C) ? (2 * n + (q >> 1)) : q;
          ls[lg] += s; lq[lg] += sq;
        }
        if constexpr (C::OUT_ESZ == 4) {
          if (pvalid) *reinterpret_cast<float4*>(out_b + act_offset(C::COUT, h, w, 0, co, gy, gx) * 4) = make_float4(v[0], v[1], v[2], v[3]);
        } else if constexpr (EK == EK_BF16) {
          pk[q].x = f32_to_bf16(v[0]) | (f32_to_bf16(v[1]) << 16);
          pk[q].y = f32_to_bf16(v[2]) | (f32_to_bf16(v[3]) << 16);
        } else {
          pk[q].x = f32_to_f16(v[0]) | (f32_to_f16(v[1]) << 16);
          pk[q].y = f32_to_f16(v[2]) | (f32_to_f16(v[3]) << 16);
        }
      }
      if constexpr (C::OUT_ESZ == 2) {
        // half-wave g holds couts 8q+4g..+3: swap so that g=0 lanes own couts 16k..16k+7, g=1 lanes 16k+8..16k+15
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const auto rx = __builtin_amdgcn_permlane32_swap(pk[2 * k].x, pk[2 * k + 1].x, false, false);
          const auto ry = __builtin_amdgcn_permlane32_swap(pk[2 * k].y, pk[2 * k + 1].y, false, false);
          // after the swap: lanes g=0: (rx[0], ry[0]) = own quad 2k, (rx[1], ry[1]) = partner's quad 2k   -> couts 16k .. 16k+7
          //                 lanes g=1: (rx[0], ry[0]) = partner's quad 2k+1, (rx[1], ry[1]) = own 2k+1     -> couts 16k+8 .. 16k+15
          if (pvalid) {
            const int co = n0 + n * 32 + 16 * k + 8 * g;
            *reinterpret_cast<uint4*>(out_b + act_offset(C::COUT, h, w, 0, co, gy, gx) * 2) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
          }
        }
      }
    }
  }
  double ds[4], dq[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { ds[k] = (double)ls[k]; dq[k] = (double)lq[k]; }
  constexpr int TOP = (C::COUT < 32) ? 16 : 32;
#pragma unroll
  for (int off = TOP; off >= 1; off >>= 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { ds[k] += __shfl_xor(ds[k], off, 64); dq[k] += __shfl_xor(dq[k], off, 64); }
  }
  double* s_red = reinterpret_cast<double*>(s_patch);               // all LDS tile reads are behind the last barrier
  if constexpr (C::COUT < 32) {
    if (li == 0) {
      s_red[wave * 8 + (0 + g) * 2 + 0] = ds[0]; s_red[wave * 8 + (0 + g) * 2 + 1] = dq[0];
      s_red[wave * 8 + (2 + g) * 2 + 0] = ds[1]; s_red[wave * 8 + (2 + g) * 2 + 1] = dq[1];
    }
  } else {
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { s_red[wave * 8 + k * 2] = ds[k]; s_red[wave * 8 + k * 2 + 1] = dq[k]; }
    }
  }
  __syncthreads();
  if (tid < 2 * NG_LOCAL) {
    double tot = 0.0;
#pragma unroll
    for (int wv = 0; wv < C::WAVES; ++wv) tot += s_red[wv * 8 + tid];
    const int gbase = (C::COUT == COND_C) ? (n0 / (COND_C / GN_GROUPS)) : 0;
    double* dst = p.stats_out + ((size_t)b * STAT_SLOTS + (blockIdx.x % STAT_SLOTS)) * STAT_STRIDE + gbase * 2 + tid;
    atomicAdd(dst, tot);
  }
}

// ------------------------------------------------------------------------------------------------
template <int EK, int LAYER>
static hipError_t launch_one2(const ConvParams& p, hipStream_t s) {
  using C = Cfg2<EK, LAYER>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm2_kernel<C>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  dim3 grid(p.tiles_x * p.tiles_y * p.B, C::COUT_PAD / C::NT);
  hipLaunchKernelGGL(conv_igemm2_kernel<C>, grid, dim3(C::THREADS), C::SMEM_BYTES, s, p);
  return hipGetLastError();
}
template <int EK>
static hipError_t launch_layer2(int layer, const ConvParams& p, hipStream_t s) {
  switch (layer) {
    case 1: return launch_one2<EK, 1>(p, s);
    case 2: return launch_one2<EK, 2>(p, s);
    case 3: return launch_one2<EK, 3>(p, s);
    case 4: return launch_one2<EK, 4>(p, s);
    default: return hipErrorInvalidValue;
  }
}
hipError_t launch_conv_igemm2(int layer, int ek, const ConvParams& p, hipStream_t s) {
  switch (ek) {
    case EK_F32: return launch_layer2<EK_F32>(layer, p, s);
    case EK_BF16: return launch_layer2<EK_BF16>(layer, p, s);
    case EK_F16: return launch_layer2<EK_F16>(layer, p, s);
    default: return hipErrorInvalidValue;
  }
}

template <int EK, int LAYER> static PackGeom geom2_of() {
  using C = Cfg2<EK, LAYER>;
  return PackGeom{C::CIN, C::COUT, C::COUT_PAD, C::CK, C::TG, C::NT};
}
template <int EK> static PackGeom geom2_layer(int layer) {
  switch (layer) {
    case 1: return geom2_of<EK, 1>();
    case 2: return geom2_of<EK, 2>();
    case 3: return geom2_of<EK, 3>();
    default: return geom2_of<EK, 4>();
  }
}
PackGeom conv_pack_geom2(int layer, int ek) {
  switch (ek) {
    case EK_F32: return geom2_layer<EK_F32>(layer);
    case EK_BF16: return geom2_layer<EK_BF16>(layer);
    default: return geom2_layer<EK_F16>(layer);
  }
}

}  // namespace dd
