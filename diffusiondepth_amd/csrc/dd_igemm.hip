// dd_igemm.hip -- fused 3x3 convolution for the epsilon-network of DiffusionDepth on gfx950 (CDNA4).
//
// One kernel template covers the four convolutions of ScheduledCNNRefine
// (reference src/model/head/ddim_depth_estimate_res.py:303-322):
//     conv1 16->64, conv2 64->256, conv3 256->64, conv4 64->16, each followed by GroupNorm(4)+ReLU.
// GroupNorm needs statistics over the whole (C/4, H, W) extent of a sample, so a layer's output must
// be complete before the next layer can normalise it.  Each launch therefore
//   prologue : applies the PREVIOUS layer's GroupNorm affine + ReLU (and, for conv3, the skip-add
//              of the condition map and the time embedding, ...res.py:330-340; for conv1 the DDIM
//              update x <- c1*x + c2*eps of the previous step, scheduling_ddim.py:285-326) while
//              staging its halo'd input tile into LDS,
//   main loop: implicit GEMM on the matrix cores.  The input tile (TH+2)x(TW+2) pixels x CK channels
//              is staged ONCE per channel chunk and re-read from LDS for the 9 taps (so HBM/L2 sees
//              the input ~1.3x, not 9x); weights stream through LDS per tap group,
//   epilogue : adds the bias, accumulates this layer's GroupNorm sum / sum-of-squares (fp32 per
//              lane -> fp64 across lanes, waves and workgroups) and stores the raw output (channel-blocked NHWC, dd_elem.h).
//
// MFMA mapping (wave64): D[cout][pixel] += W[cout][k] * P[k][pixel] with the WEIGHTS as the A operand
// and the PIXELS as the B operand of v_mfma_f32_32x32x16_{bf16,f16} (K = 16 channels) or of 4 x
// v_mfma_f32_32x32x2_f32 (K = 8 channels, exact fp32).  Lane l supplies 16 bytes of channels
// [.. + 8*(l>>5)) of cout/pixel (l&31) from LDS rows padded by 16 B (row strides 48/80/144 B ->
// conflict-free ds_read_b128).  In D a lane owns ONE pixel and 4 consecutive couts per register
// quad, so the NHWC store and all per-pixel epilogue math are lane-local.
#include "dd_elem.h"
#include "dd_gcn.h"

namespace dd {

// ------------------------------------------------------------------------------------------------
// per-layer configuration
// ------------------------------------------------------------------------------------------------
template <int EK_, int LAYER_> struct Cfg {
  static constexpr int EK = EK_;
  static constexpr int LAYER = LAYER_;
  static constexpr int ESZ = ElemSize<EK>::V;
  static constexpr int CIN = (LAYER == 1) ? LATENT_C : (LAYER == 2) ? HID_C : (LAYER == 3) ? COND_C : HID_C;
  static constexpr int COUT = (LAYER == 1) ? HID_C : (LAYER == 2) ? COND_C : (LAYER == 3) ? HID_C : LATENT_C;
  static constexpr int COUT_PAD = (COUT < 32) ? 32 : COUT;
  // channels staged per chunk: 64 B (conv1: all 16 ch -> 32/64 B), 128 B for conv2/conv4, 64 B for conv3
  static constexpr int CK = (LAYER == 1) ? 16 : (LAYER == 3) ? (64 / ESZ) : (128 / ESZ);
  static constexpr int TG = (LAYER == 1) ? 9 : (LAYER == 2) ? 1 : 3;      // taps per weight stage
  static constexpr int NT = (LAYER == 2) ? 128 : COUT_PAD;               // couts per workgroup
  static constexpr int TH = 8, TW = 32;                                  // output tile (pixels)
  static constexpr int WAVES = 4;                                        // 4 x 1 over (pixels, couts)
  static constexpr int THREADS = WAVES * 64;
  static constexpr int WM = (TH * TW) / (32 * WAVES);                    // 32-pixel blocks per wave (2)
  static constexpr int WN = NT / 32;                                     // 32-cout blocks per wave
  static constexpr int PRO = (LAYER == 1) ? PRO_X : (LAYER == 3) ? PRO_GN_ADD : PRO_GN;
  static constexpr int IN_ESZ = (LAYER == 1) ? 4 : ESZ;                  // conv1 reads the fp32 state
  static constexpr int OUT_ESZ = (LAYER == 4) ? 4 : ESZ;                 // conv4 writes fp32
  static constexpr int PH = TH + 2, PW = TW + 2;
  static constexpr int ROWB = CK * ESZ;
  static constexpr int PSTR = ROWB + 16;                                 // padded LDS row stride (48/80/144 B)
  static constexpr int PPP = ROWB / 16;                                  // 16-B pieces per row
  static constexpr int EPP = 16 / ESZ;                                   // elements per piece
  static constexpr int NCHUNK = CIN / CK;
  static constexpr int NTG = 9 / TG;
  static constexpr int PATCH_BYTES = PH * PW * PSTR;
  static constexpr int W_BYTES = TG * NT * PSTR;
  static constexpr int CTAB = (LAYER == 1) ? LATENT_C : CIN;             // channels the prologue normalises
  static constexpr int TAB_FLOATS = 3 * CTAB + 8;
  static constexpr int SMEM_BYTES = PATCH_BYTES + W_BYTES + TAB_FLOATS * 4;
  static_assert(CIN % CK == 0 && 9 % TG == 0 && COUT_PAD % NT == 0, "tiling");
  static_assert(PATCH_BYTES % 16 == 0 && W_BYTES % 16 == 0, "LDS carve alignment");
  static_assert(PATCH_BYTES >= STAT_SLOTS * 8 * 8 + 64 + WAVES * 8 * 8, "scratch fits in the patch region");
};

template <class C>
__global__ void __launch_bounds__(C::THREADS) conv_igemm_kernel(ConvParams p) {
  constexpr int EK = C::EK;
  constexpr int PW = C::PW, PSTR = C::PSTR, PPP = C::PPP, EPP = C::EPP, CK = C::CK;
  DD_DYN_SMEM(smem);
  char* s_patch = smem;
  char* s_w = smem + C::PATCH_BYTES;
  float* s_tab = reinterpret_cast<float*>(smem + C::PATCH_BYTES + C::W_BYTES);
  float* tab_a = s_tab;
  float* tab_b = s_tab + C::CTAB;
  float* tab_e = s_tab + 2 * C::CTAB;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, g = lane >> 5;

  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const int b = blockIdx.x / tiles_per_img;
  const int trem = blockIdx.x - b * tiles_per_img;
  const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
  const int y0 = ty * C::TH, x0 = tx * C::TW;
  const int n0 = blockIdx.y * C::NT;
  const int h = p.h, w = p.w;

  // ---- prologue table: GroupNorm affine of the producing layer ---------------------------------
  const bool have_norm = (C::PRO != PRO_X) || (p.step > 0);
  float c1 = 1.f, c2 = 0.f;
  if (have_norm) {
    double* s_tmp = reinterpret_cast<double*>(s_patch);           // [STAT_SLOTS][8] then [8] sums
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
      double var = s_tmp[STAT_SLOTS * 8 + grp * 2 + 1] / cnt - mean * mean;   // biased, as torch
      var = var > 0.0 ? var : 0.0;
      const double rstd = 1.0 / sqrt(var + (double)GN_EPS);
      const double a = (double)p.gn_gamma[c] * rstd;
      tab_a[c] = (float)a;
      tab_b[c] = (float)((double)p.gn_beta[c] - mean * a);
      if constexpr (C::PRO == PRO_GN_ADD) {
        const long long t = clamp_t(p.tvec[p.t_base + b * p.t_bstride]);
        tab_e[c] = p.emb[(size_t)t * COND_C + c];
      }
    }
    if constexpr (C::PRO == PRO_X) { c1 = p.c1c2[2 * (p.step - 1)]; c2 = p.c1c2[2 * (p.step - 1) + 1]; }
  }

  f32x16_t acc[C::WN][C::WM];
#pragma unroll
  for (int n = 0; n < C::WN; ++n)
#pragma unroll
    for (int m = 0; m < C::WM; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[n][m][r] = 0.f;

  // per-lane LDS bases: pixel fragment rows (tap (0,0) = top-left of the 3x3 window) and weight rows
  int pbase[C::WM];
#pragma unroll
  for (int m = 0; m < C::WM; ++m) {
    const int pix = (wave * C::WM + m) * 32 + li;
    const int r = pix / C::TW, c = pix - r * C::TW;
    pbase[m] = (r * PW + c) * PSTR + g * 16;
  }
  const int wbase = li * PSTR + g * 16;

  constexpr int IN_ESZ = C::IN_ESZ;
  constexpr int NLD = EPP * IN_ESZ / 16;                          // uint4 loads per item from the input
  const char* in_b = reinterpret_cast<const char*>(p.in) + (size_t)b * h * w * C::CIN * IN_ESZ;
  const char* cond_b = (C::PRO == PRO_GN_ADD) ? reinterpret_cast<const char*>(p.cond) + (size_t)b * h * w * C::CIN * IN_ESZ : nullptr;
  const char* y4_b = (C::PRO == PRO_X) ? reinterpret_cast<const char*>(p.y4) + (size_t)b * h * w * LATENT_C * 4 : nullptr;
  char* xout_b = (C::PRO == PRO_X) ? reinterpret_cast<char*>(p.xout) + (size_t)b * h * w * LATENT_C * 4 : nullptr;

  for (int chunk = 0; chunk < C::NCHUNK; ++chunk) {
    __syncthreads();                       // table visible / previous chunk's reads of the patch done
    // ---- stage the halo'd input patch for this channel chunk ------------------------------------
    constexpr int ITEMS = C::PH * PW * PPP;
    constexpr int U = 4;
    for (int base = 0; base < ITEMS; base += C::THREADS * U) {
      uint4 raw[U][NLD];
      uint4 aux[U][NLD];                   // PRO_GN_ADD: cond ; PRO_X: y4
      bool inside[U];
      int pp_[U], j_[U];
      size_t goff[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int it = base + u * C::THREADS + tid;
        const int pp = it / PPP, j = it - pp * PPP;
        const int pr = pp / PW, pc = pp - pr * PW;
        const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
        pp_[u] = pp; j_[u] = j;
        inside[u] = (it < ITEMS) && gy >= 0 && gy < h && gx >= 0 && gx < w;
        goff[u] = act_offset(C::CIN, h, w, 0, chunk * CK + j * EPP, gy, gx) * IN_ESZ;     // channel-blocked for CIN >= 32
        if (inside[u]) {
#pragma unroll
          for (int q = 0; q < NLD; ++q) raw[u][q] = *reinterpret_cast<const uint4*>(in_b + goff[u] + q * 16);
          if constexpr (C::PRO == PRO_GN_ADD) {
            aux[u][0] = *reinterpret_cast<const uint4*>(cond_b + goff[u]);
          } else if constexpr (C::PRO == PRO_X) {
            if (have_norm) {
#pragma unroll
              for (int q = 0; q < NLD; ++q) aux[u][q] = *reinterpret_cast<const uint4*>(y4_b + goff[u] + q * 16);
            }
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int it = base + u * C::THREADS + tid;
        if (it >= ITEMS) continue;
        float v[EPP];
        if (inside[u]) {
          const int c0 = (C::PRO == PRO_X) ? j_[u] * EPP : chunk * CK + j_[u] * EPP;
          if constexpr (C::PRO == PRO_X) {
            // DDIM update of the previous step fused into the load: x <- c1*x + c2*relu(gn4(y4))
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
            const int pr = pp_[u] / PW, pc = pp_[u] - pr * PW;
            if (have_norm && pr >= 1 && pr <= C::TH && pc >= 1 && pc <= C::TW) {
#pragma unroll
              for (int q = 0; q < NLD; ++q)
                *reinterpret_cast<float4*>(xout_b + goff[u] + q * 16) = make_float4(v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]);
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
          for (int i = 0; i < EPP; ++i) v[i] = 0.f;     // conv zero padding applies AFTER the normalisation
        }
        *reinterpret_cast<uint4*>(s_patch + pp_[u] * PSTR + j_[u] * 16) = Piece<EK>::pack(v);
      }
    }

    // ---- tap groups: stream weights through LDS, contract on the matrix cores -------------------
    for (int tg = 0; tg < C::NTG; ++tg) {
      if (tg > 0) __syncthreads();         // previous tap group's weight reads done
      {
        constexpr int WPIECES = C::TG * C::NT * PPP;
        const char* wsrc = reinterpret_cast<const char*>(p.wpack) +
                           ((size_t)((blockIdx.y * C::NCHUNK + chunk) * C::NTG + tg)) * (size_t)(C::TG * C::NT * C::ROWB);
        for (int q = tid; q < WPIECES; q += C::THREADS) {
          const int row = q / PPP, j = q - row * PPP;
          *reinterpret_cast<uint4*>(s_w + row * PSTR + j * 16) = *reinterpret_cast<const uint4*>(wsrc + (size_t)q * 16);
        }
      }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < C::TG; ++t) {
        const int tap = tg * C::TG + t;
        const int dy = tap / 3, dx = tap - dy * 3;
        const int poff = (dy * PW + dx) * PSTR;
#pragma unroll
        for (int kq = 0; kq < PPP / 2; ++kq) {
          uint4 pf[C::WM], wf[C::WN];
#pragma unroll
          for (int m = 0; m < C::WM; ++m) pf[m] = *reinterpret_cast<const uint4*>(s_patch + pbase[m] + poff + kq * 32);
#pragma unroll
          for (int n = 0; n < C::WN; ++n) wf[n] = *reinterpret_cast<const uint4*>(s_w + wbase + (t * C::NT + n * 32) * PSTR + kq * 32);
#pragma unroll
          for (int n = 0; n < C::WN; ++n)
#pragma unroll
            for (int m = 0; m < C::WM; ++m) mma_step<EK>(acc[n][m], wf[n], pf[m]);
        }
      }
    }
  }

  // ---- epilogue: bias, GroupNorm partial sums, NHWC store -----------------------------------------
  constexpr int NG_LOCAL = (C::COUT == COND_C) ? C::NT / (COND_C / GN_GROUPS) : 4;   // groups this WG touches
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
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cl = n * 32 + 8 * q + 4 * g;          // cout within this workgroup's NT tile
        const int co = n0 + cl;
        if (C::COUT < 32 && q >= 2) continue;            // conv4: couts 16..31 are zero padding
        const float4 bv = *reinterpret_cast<const float4*>(p.bias + co);
        float v[4] = {acc[n][m][q * 4 + 0] + bv.x, acc[n][m][q * 4 + 1] + bv.y,
                      acc[n][m][q * 4 + 2] + bv.z, acc[n][m][q * 4 + 3] + bv.w};
        if (pvalid) {
          const float s = (v[0] + v[1]) + (v[2] + v[3]);
          const float sq = (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
          // local group index: COUT=256 -> 64-cout groups (n/2); COUT=64 -> 16-cout groups (2n + q/2);
          // COUT=16 -> 4-cout groups: index q here, lane half g resolved after the reduction
          constexpr int dummy = 0; (void)dummy;
          const int lg = (C::COUT == COND_C) ? (n >> 1) : (C::COUT == HID_C) ? (2 * n + (q >> 1)) : q;
          ls[lg] += s; lq[lg] += sq;
          if constexpr (C::OUT_ESZ == 4) {
            *reinterpret_cast<float4*>(out_b + act_offset(C::COUT, h, w, 0, co, gy, gx) * 4) = make_float4(v[0], v[1], v[2], v[3]);
          } else {
            uint2 pk;
            if constexpr (EK == EK_BF16) {
              pk.x = f32_to_bf16(v[0]) | (f32_to_bf16(v[1]) << 16);
              pk.y = f32_to_bf16(v[2]) | (f32_to_bf16(v[3]) << 16);
            } else {
              pk.x = f32_to_f16(v[0]) | (f32_to_f16(v[1]) << 16);
              pk.y = f32_to_f16(v[2]) | (f32_to_f16(v[3]) << 16);
            }
            *reinterpret_cast<uint2*>(out_b + act_offset(C::COUT, h, w, 0, co, gy, gx) * 2) = pk;
          }
        }
      }
    }
  }
  // cross-lane reduction in fp64.  COUT=16: lanes of half g hold groups {g, 2+g} -> reduce within
  // each 32-lane half only; otherwise over the full wave.
  double ds[4], dq[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { ds[k] = (double)ls[k]; dq[k] = (double)lq[k]; }
  constexpr int TOP = (C::COUT < 32) ? 16 : 32;
#pragma unroll
  for (int off = TOP; off >= 1; off >>= 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { ds[k] += __shfl_xor(ds[k], off, 64); dq[k] += __shfl_xor(dq[k], off, 64); }
  }
  __syncthreads();                         // all waves finished reading the LDS tiles
  double* s_red = reinterpret_cast<double*>(s_patch);               // [WAVES][8]
  if constexpr (C::COUT < 32) {
    if (li == 0) {                         // lane 0: groups 0 (q=0), 2 (q=1); lane 32: groups 1, 3
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
// host side
// ------------------------------------------------------------------------------------------------
template <int EK, int LAYER>
static hipError_t launch_one(const ConvParams& p, hipStream_t s) {
  using C = Cfg<EK, LAYER>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm_kernel<C>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  dim3 grid(p.tiles_x * p.tiles_y * p.B, C::COUT_PAD / C::NT);
  hipLaunchKernelGGL(conv_igemm_kernel<C>, grid, dim3(C::THREADS), C::SMEM_BYTES, s, p);
  return hipGetLastError();
}

template <int EK>
static hipError_t launch_layer(int layer, const ConvParams& p, hipStream_t s) {
  switch (layer) {
    case 1: return launch_one<EK, 1>(p, s);
    case 2: return launch_one<EK, 2>(p, s);
    case 3: return launch_one<EK, 3>(p, s);
    case 4: return launch_one<EK, 4>(p, s);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_conv_igemm(int layer, int ek, const ConvParams& p, hipStream_t s) {
  switch (ek) {
    case EK_F32: return launch_layer<EK_F32>(layer, p, s);
    case EK_BF16: return launch_layer<EK_BF16>(layer, p, s);
    case EK_F16: return launch_layer<EK_F16>(layer, p, s);
    default: return hipErrorInvalidValue;
  }
}

template <int EK, int LAYER> static PackGeom geom_of() {
  using C = Cfg<EK, LAYER>;
  return PackGeom{C::CIN, C::COUT, C::COUT_PAD, C::CK, C::TG, C::NT, C::TH, 3};
}
template <int EK> static PackGeom geom_layer(int layer) {
  switch (layer) {
    case 1: return geom_of<EK, 1>();
    case 2: return geom_of<EK, 2>();
    case 3: return geom_of<EK, 3>();
    default: return geom_of<EK, 4>();
  }
}
PackGeom conv_pack_geom(int layer, int ek) {
  switch (ek) {
    case EK_F32: return geom_layer<EK_F32>(layer);
    case EK_BF16: return geom_layer<EK_BF16>(layer);
    default: return geom_layer<EK_F16>(layer);
  }
}

}  // namespace dd
