// dd_igemm2ws.hip -- wave-specialised variant of the v2 fused convolution for the multi-chunk 256->64 layers
// (conv3 of the Res denoiser, pred.0 of the Swin variant, hoisted conv3).
//
// Same tiling, LDS images, swizzles and epilogue as dd_igemm2.hip, but a workgroup has 8 waves in two roles:
//   waves 0..3  MFMA role   : ds_read fragments + MFMA + ONE barrier per (chunk, tap-group) stage, then the epilogue;
//   waves 4..7  staging role: weight LDS-DMA one stage ahead, raw patch loads of the next channel chunk, and the
//                             GroupNorm / ReLU (/ + cond + E[t]) transform into the other patch buffer.
// The two roles are separate loop nests (so that the accumulators and the raw-patch registers are never live in the
// same code and both fit 128 VGPRs -> 16 waves = two workgroups per CU) that execute the same number of barriers.
// In dd_igemm2.hip every wave stops issuing MFMAs while it normalises the next chunk (VALU burst before the chunk's
// last barrier); here that work runs on the SIMDs' VALU while the MFMA waves keep the matrix pipe busy.
#define DD_FAT_CONV3 0      // the wave-specialised variant keeps the 8x32 / 3-taps-per-stage tiling
#include "dd_igemm2_cfg.h"
#include "dd_gcn.h"

namespace dd {

template <class C>
__global__ void __launch_bounds__(512, 4) conv_igemm2ws_kernel(ConvParams p) {
  constexpr int EK = C::EK;
  constexpr int PW = C::PW, ROWB = C::ROWB, PPP = C::PPP, RPB = C::RPB, EPP = C::EPP, CK = C::CK;
  constexpr int IN_ESZ = C::IN_ESZ, NKQ = C::NKQ, NLD = C::NLD;
  constexpr int STHREADS = 256;
  constexpr int NIT = (C::ITEMS + STHREADS - 1) / STHREADS;
  static_assert(C::WAVES == 4 && C::NCHUNK > 1 && C::NPB == 2 && C::NWB == 2 && C::TG == 3, "layers this variant is written for");
  static_assert(C::PRO != PRO_X && NLD == 1, "PRO_GN / PRO_GN_ADD / PRO_RAW inputs");
  DD_DYN_SMEM(smem);
  float* s_tab = reinterpret_cast<float*>(smem + C::NPB * C::PATCH_BYTES + C::NWB * C::W_BYTES);
  float* tab_a = s_tab;
  float* tab_b = s_tab + C::CTAB;
  float* tab_e = s_tab + 2 * C::CTAB;
  float* tab_bias = s_tab + 3 * C::CTAB;
  float* tab_et = tab_bias + C::NT;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wave_all & 3;                 // index within the role
  const int li = lane & 31, g = lane >> 5;

  constexpr int NSPLIT = C::COUT_PAD / C::NT;
  int wgid;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int nsplit = wgid % NSPLIT;
  const int tile = wgid / NSPLIT;
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const int b = tile / tiles_per_img;
  const int trem = tile - b * tiles_per_img;
  const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
  const int y0 = ty * C::TH, x0 = tx * C::TW;
  const int n0 = nsplit * C::NT;
  const int h = p.h, w = p.w;
  constexpr bool have_norm = (C::PRO != PRO_RAW);
  constexpr int NRAW = NIT * ((C::PRO == PRO_GN_ADD) ? 2 : 1);
  static_assert(NRAW <= 63, "vmcnt field");

  if (wave_all >= 4) {
    // =============================== staging role ===========================================================
    const int stid = tid & (STHREADS - 1);
    const char* in_b = reinterpret_cast<const char*>(p.in) + (size_t)b * h * w * C::CIN * IN_ESZ;
    const char* cond_b = (C::PRO == PRO_GN_ADD) ? reinterpret_cast<const char*>(p.cond) + (size_t)b * h * w * C::CIN * IN_ESZ : nullptr;
    const unsigned lds_base = DD_LDS_BASE(smem);
    auto issue_weights = [&](int s) {
      const char* src = reinterpret_cast<const char*>(p.wpack) + ((size_t)nsplit * C::NSTAGE + s) * (size_t)C::W_BYTES + lane * 16;
      const unsigned dst = lds_base + C::W_OFF + (s & 1) * C::W_BYTES;
#pragma unroll
      for (int c = 0; c < (C::W_BYTES / 1024 + 3) / 4; ++c) {
        const int kc = c * 4 + wave;
        if (kc < C::W_BYTES / 1024) {
          const char* gsrc = src + (size_t)kc * 1024;
          const unsigned ldst = __builtin_amdgcn_readfirstlane(dst + kc * 1024);
          DD_LDS_DMA16(smem, gsrc, ldst);
        }
      }
    };
    const int jfix = stid & (PPP - 1);
    int lds_off[NIT], pix_off[NIT];
    unsigned m_valid = 0, m_inside = 0;
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      const int it = u * STHREADS + stid;
      const int itc = it < C::ITEMS ? it : C::ITEMS - 1;
      const int pp = itc >> C::LOG2_PPP;
      const int pr = pp / PW, pc = pp - pr * PW;
      const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
      if (it < C::ITEMS) m_valid |= 1u << u;
      if (gy >= 0 && gy < h && gx >= 0 && gx < w) m_inside |= 1u << u;
      const int gyc = gy < 0 ? 0 : (gy >= h ? h - 1 : gy);
      const int gxc = gx < 0 ? 0 : (gx >= w ? w - 1 : gx);
      pix_off[u] = gyc * w + gxc;
      lds_off[u] = pp * ROWB + ((jfix << 4) ^ swz16<RPB, PPP>(pc));
    }
    uint4 raw[NIT], aux[NIT];
    auto load_raw = [&](int chunk) {
      const int cbase = chunk * CK + jfix * EPP;
      const size_t off0 = (size_t)(cbase >> 5) * h * w * ACT_CB + (cbase & (ACT_CB - 1));
#pragma unroll
      for (int u = 0; u < NIT; ++u) {
        const size_t goff = (off0 + (size_t)pix_off[u] * ACT_CB) * IN_ESZ;
        raw[u] = *reinterpret_cast<const uint4*>(in_b + goff);
        if constexpr (C::PRO == PRO_GN_ADD) aux[u] = *reinterpret_cast<const uint4*>(cond_b + goff);
      }
      DD_VMEM_LOADS_ISSUED(NIT * (C::PRO == PRO_GN_ADD ? 2 : 1));      // host model only (dd_gcn.h)
    };
    auto transform_write = [&](int chunk, int pbuf_off) {
      float ta[EPP], tb[EPP], te[EPP];
      const int c0 = chunk * CK + jfix * EPP;
      if constexpr (have_norm) {
#pragma unroll
        for (int q = 0; q < EPP / 4; ++q) {
          const float4 a4 = *reinterpret_cast<const float4*>(tab_a + c0 + 4 * q);
          const float4 b4 = *reinterpret_cast<const float4*>(tab_b + c0 + 4 * q);
          ta[4 * q] = a4.x; ta[4 * q + 1] = a4.y; ta[4 * q + 2] = a4.z; ta[4 * q + 3] = a4.w;
          tb[4 * q] = b4.x; tb[4 * q + 1] = b4.y; tb[4 * q + 2] = b4.z; tb[4 * q + 3] = b4.w;
          if constexpr (C::PRO == PRO_GN_ADD) {
            const float4 e4 = *reinterpret_cast<const float4*>(tab_e + c0 + 4 * q);
            te[4 * q] = e4.x; te[4 * q + 1] = e4.y; te[4 * q + 2] = e4.z; te[4 * q + 3] = e4.w;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < NIT; ++u) {
        if (!((m_valid >> u) & 1u)) continue;
        uint4 outv = make_uint4(0u, 0u, 0u, 0u);       // zero padding applies AFTER the normalisation
        if ((m_inside >> u) & 1u) {
          if constexpr (C::PRO == PRO_RAW) {
            outv = raw[u];
          } else {
            float v[EPP];
            Piece<EK>::unpack(raw[u], v);
#pragma unroll
            for (int i = 0; i < EPP; ++i) v[i] = fmaxf(fmaf(ta[i], v[i], tb[i]), 0.f);
            if constexpr (C::PRO == PRO_GN_ADD) {
              float cv[EPP];
              Piece<EK>::unpack(aux[u], cv);
#pragma unroll
              for (int i = 0; i < EPP; ++i) v[i] = v[i] + (cv[i] + te[i]);
            }
            outv = Piece<EK>::pack(v);
          }
        }
        *reinterpret_cast<uint4*>(smem + pbuf_off + lds_off[u]) = outv;
      }
    };

    issue_weights(0);
    load_raw(0);
    DD_WAIT_LGKM0();
    __builtin_amdgcn_s_barrier();                       // #1: GroupNorm table (built by the MFMA role) is visible
    asm volatile("" ::: "memory");
    transform_write(0, 0);
    DD_WAIT_VM_LGKM0(0);
    __builtin_amdgcn_s_barrier();                       // #2: patch 0 and weight stage 0 are in LDS
    asm volatile("" ::: "memory");
#pragma unroll 1
    for (int chunk = 0; chunk < C::NCHUNK; ++chunk) {
#pragma unroll 1
      for (int tg = 0; tg < C::NTG; ++tg) {
        const int s = chunk * C::NTG + tg;
        if (s + 1 < C::NSTAGE) issue_weights(s + 1);
        asm volatile("" ::: "memory");                  // DMA stays ahead of the raw loads in issue order
        const bool next = chunk + 1 < C::NCHUNK;
        if (tg == 0 && next) load_raw(chunk + 1);
        if (tg == 1 && next) transform_write(chunk + 1, ((chunk + 1) & 1) * C::PATCH_BYTES);
        // the DMA of stage s+1 must have landed before the barrier; raw loads issued in THIS stage may keep flying
        if (tg == 0 && next) DD_WAIT_VM_LGKM0(NRAW);
        else DD_WAIT_VM_LGKM0(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
    return;
  }

  // ================================= MFMA role ================================================================
  if (tid < C::NT) tab_bias[tid] = p.bias[n0 + tid];
  if constexpr (C::ADD_C) {
    const long long t = clamp_t(p.tvec[p.t_base + b * p.t_bstride]);
    for (int i = tid; i < 10 * HID_C; i += 256) tab_et[i] = p.etab[(size_t)t * 10 * HID_C + i];
  }
  if constexpr (have_norm) {
    static_assert(C::PRO == PRO_RAW || C::CTAB == 256, "one table channel per MFMA-role thread");
    const double* st = p.stats_in + (size_t)b * STAT_SLOTS * STAT_STRIDE + (lane >> 1) * STAT_STRIDE + (lane & 1) * 4;
    double2 sv0 = *reinterpret_cast<const double2*>(st);
    double2 sv1 = *reinterpret_cast<const double2*>(st + 2);
    const float my_gamma = p.gn_gamma[tid], my_beta = p.gn_beta[tid];
    float my_emb = 0.f;
    if constexpr (C::PRO == PRO_GN_ADD) {
      const long long t = clamp_t(p.tvec[p.t_base + b * p.t_bstride]);
      my_emb = p.emb[(size_t)t * COND_C + tid];
    }
#pragma unroll
    for (int off = 2; off <= 32; off <<= 1) {
      sv0.x += __shfl_xor(sv0.x, off, 64); sv0.y += __shfl_xor(sv0.y, off, 64);
      sv1.x += __shfl_xor(sv1.x, off, 64); sv1.y += __shfl_xor(sv1.y, off, 64);
    }
    const double2 ov0 = make_double2(__shfl_xor(sv0.x, 1, 64), __shfl_xor(sv0.y, 1, 64));
    const double2 ov1 = make_double2(__shfl_xor(sv1.x, 1, 64), __shfl_xor(sv1.y, 1, 64));
    const bool hi = lane & 1;
    const double2 g0 = hi ? ov0 : sv0, g1 = hi ? ov1 : sv1, g2 = hi ? sv0 : ov0, g3 = hi ? sv1 : ov1;
    constexpr int CG = (C::CTAB > 0) ? C::CTAB / GN_GROUPS : 1;
    const int grp = tid / CG;
    const double2 gs = grp == 0 ? g0 : grp == 1 ? g1 : grp == 2 ? g2 : g3;
    const double inv_cnt = 1.0 / ((double)h * (double)w * (double)CG);
    const double mean = gs.x * inv_cnt;
    double var = gs.y * inv_cnt - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double a = (double)my_gamma / sqrt(var + (double)GN_EPS);
    tab_a[tid] = (float)a;
    tab_b[tid] = (float)((double)my_beta - mean * a);
    if constexpr (C::PRO == PRO_GN_ADD) tab_e[tid] = my_emb;
  }
  DD_WAIT_VM_LGKM0(0);
  __builtin_amdgcn_s_barrier();                         // #1
  asm volatile("" ::: "memory");
  if constexpr (C::ADD_C) {
    if (tid < HID_C) tab_bias[tid] += tab_et[9 * HID_C + tid];
  }

  f32x16_t acc[C::WN][C::WM];
#pragma unroll
  for (int m = 0; m < C::WM; ++m) {
    const int gy = y0 + wave * C::WM + m, gx = x0 + li;
    const bool pv = gy < h && gx < w;
#pragma unroll
    for (int n = 0; n < C::WN; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (C::ADD_C) {
          (void)pv;   // accumulator-fragment order written by layer 8 (dd_igemm2.hip), 4 MFMA-role waves per tile
          cv = reinterpret_cast<const float4*>(p.cadd)[((((size_t)tile * 4 + wave) * C::WN + n) * C::WM + m) * 256 + q * 64 + lane];
        }
        acc[n][m][q * 4 + 0] = cv.x; acc[n][m][q * 4 + 1] = cv.y; acc[n][m][q * 4 + 2] = cv.z; acc[n][m][q * 4 + 3] = cv.w;
      }
  }
  const int g16 = g << 4;
  int colt[3][NKQ];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int kq = 0; kq < NKQ; ++kq)
      colt[dx][kq] = wave * (C::WM * PW * ROWB) + (li + dx) * ROWB + (((kq << 5) | g16) ^ swz16<RPB, PPP>(li + dx));
  int wkt[NKQ];
#pragma unroll
  for (int kq = 0; kq < NKQ; ++kq) wkt[kq] = C::W_OFF + li * ROWB + (((kq << 5) | g16) ^ swz16<RPB, PPP>(li));
  DD_WAIT_VM_LGKM0(0);
  __builtin_amdgcn_s_barrier();                         // #2
  asm volatile("" ::: "memory");

#pragma unroll 1
  for (int chunk = 0; chunk < C::NCHUNK; ++chunk) {
#pragma unroll 1
    for (int tg = 0; tg < C::NTG; ++tg) {
      const int s = chunk * C::NTG + tg;
      const int poff = (chunk & 1) * C::PATCH_BYTES;
      const int woff = (s & 1) * C::W_BYTES;
      int wa[NKQ];
#pragma unroll
      for (int kq = 0; kq < NKQ; ++kq) wa[kq] = wkt[kq] + woff;
#pragma unroll
      for (int t = 0; t < C::TG; ++t) {                 // TG == 3: dy = tg, dx = t
        const int roff = poff + tg * (PW * ROWB);
#pragma unroll
        for (int kq = 0; kq < NKQ; ++kq) {
          const int pa = colt[t][kq] + roff;
          uint4 pf[C::WM], wf[C::WN];
#pragma unroll
          for (int m = 0; m < C::WM; ++m) pf[m] = *reinterpret_cast<const uint4*>(smem + pa + m * (PW * ROWB));
#pragma unroll
          for (int n = 0; n < C::WN; ++n) wf[n] = *reinterpret_cast<const uint4*>(smem + wa[kq] + (t * C::NT + n * 32) * ROWB);
#pragma unroll
          for (int n = 0; n < C::WN; ++n)
#pragma unroll
            for (int m = 0; m < C::WM; ++m) mma_step<EK>(acc[n][m], wf[n], pf[m]);
        }
      }
      DD_WAIT_LGKM0();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
  }

  // ---- epilogue (identical to dd_igemm2.hip; only the four MFMA waves are alive) ---------------------------------------
  float ls[4] = {0.f, 0.f, 0.f, 0.f}, lq[4] = {0.f, 0.f, 0.f, 0.f};
  char* out_b = reinterpret_cast<char*>(p.out) + (size_t)b * h * w * C::COUT * C::OUT_ESZ;
#pragma unroll
  for (int m = 0; m < C::WM; ++m) {
    const int gy = y0 + wave * C::WM + m, gx = x0 + li;
    const bool pvalid = gy < h && gx < w;
    unsigned tapmask = 0x1FFu;
    if constexpr (C::ADD_C) {
      const unsigned ry = (gy >= 1 ? 1u : 0u) | 2u | (gy + 1 < h ? 4u : 0u);
      const unsigned rx = (gx >= 1 ? 1u : 0u) | 2u | (gx + 1 < w ? 4u : 0u);
      tapmask = ((ry & 1u) ? rx : 0u) | ((ry & 2u) ? (rx << 3) : 0u) | ((ry & 4u) ? (rx << 6) : 0u);
    }
#pragma unroll
    for (int n = 0; n < C::WN; ++n) {
      uint2 pk[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int co = n0 + n * 32 + 8 * q + 4 * g;
        float4 bv = *reinterpret_cast<const float4*>(tab_bias + n * 32 + 8 * q + 4 * g);
        if constexpr (C::ADD_C) {
          if (tapmask != 0x1FFu) {
            const int cl = n * 32 + 8 * q + 4 * g;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
              if (!((tapmask >> tap) & 1u)) {
                const float4 ev = *reinterpret_cast<const float4*>(tab_et + tap * HID_C + cl);
                bv.x -= ev.x; bv.y -= ev.y; bv.z -= ev.z; bv.w -= ev.w;
              }
            }
          }
        }
        const float v[4] = {acc[n][m][q * 4 + 0] + bv.x, acc[n][m][q * 4 + 1] + bv.y,
                            acc[n][m][q * 4 + 2] + bv.z, acc[n][m][q * 4 + 3] + bv.w};
        if (C::STATS && pvalid) {
          const float s = (v[0] + v[1]) + (v[2] + v[3]);
          const float sq = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3])));
          const int lg = 2 * n + (q >> 1);               // COUT = 64: 16-cout groups
          ls[lg] += s; lq[lg] += sq;
        }
        if constexpr (C::OUT_ESZ == 4) {
          if (pvalid) *reinterpret_cast<float4*>(out_b + act_offset(C::COUT, h, w, 0, co, gy, gx) * 4) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
          pk[q].x = pack2<EK>(v[0], v[1]);
          pk[q].y = pack2<EK>(v[2], v[3]);
        }
      }
      if constexpr (C::OUT_ESZ == 2) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const auto rx = __builtin_amdgcn_permlane32_swap(pk[2 * k].x, pk[2 * k + 1].x, false, false);
          const auto ry = __builtin_amdgcn_permlane32_swap(pk[2 * k].y, pk[2 * k + 1].y, false, false);
          if (pvalid) {
            const int co = n0 + n * 32 + 16 * k + 8 * g;
            *reinterpret_cast<uint4*>(out_b + act_offset(C::COUT, h, w, 0, co, gy, gx) * 2) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
          }
        }
      }
    }
  }
  if constexpr (!C::STATS) return;
  static_assert(C::COUT == HID_C, "statistics grouping below assumes 64 couts");
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { ls[k] += __shfl_xor(ls[k], off, 64); lq[k] += __shfl_xor(lq[k], off, 64); }
  }
  double* s_red = reinterpret_cast<double*>(smem);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { s_red[wave * 8 + k * 2] = (double)ls[k]; s_red[wave * 8 + k * 2 + 1] = (double)lq[k]; }
  }
  __syncthreads();                                      // the staging waves have exited: only live waves are counted
  if (tid < 8) {
    double tot = 0.0;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) tot += s_red[wv * 8 + tid];
    atomicAdd(p.stats_out + ((size_t)b * STAT_SLOTS + (wgid % STAT_SLOTS)) * STAT_STRIDE + tid, tot);
  }
}

template <int EK, int LAYER>
static hipError_t launch_ws(const ConvParams& p, hipStream_t s) {
  using C = Cfg2<EK, LAYER>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_igemm2ws_kernel<C>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, C::SMEM_BYTES);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  dim3 grid(p.tiles_x * p.tiles_y * p.B * (C::COUT_PAD / C::NT), 1);
  hipLaunchKernelGGL(conv_igemm2ws_kernel<C>, grid, dim3(512), C::SMEM_BYTES, s, p);
  return hipGetLastError();
}

template <int EK>
static hipError_t launch_ws_layer(int layer, const ConvParams& p, hipStream_t s) {
  switch (layer) {
    case 3: return launch_ws<EK, 3>(p, s);
    case 7: return launch_ws<EK, 7>(p, s);
    case 9: return launch_ws<EK, 9>(p, s);
    default: return hipErrorInvalidValue;
  }
}

bool conv_igemm2ws_supports(int layer) { return layer == 3 || layer == 7 || layer == 9; }

hipError_t launch_conv_igemm2ws(int layer, int ek, const ConvParams& p, hipStream_t s) {
  switch (ek) {
    case EK_F32: return launch_ws_layer<EK_F32>(layer, p, s);
    case EK_BF16: return launch_ws_layer<EK_BF16>(layer, p, s);
    case EK_F16: return launch_ws_layer<EK_F16>(layer, p, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace dd
