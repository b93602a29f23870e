// dd_igemm2.hip -- the fused 3x3 convolution: implicit GEMM on MFMA (D[cout][pixel] += W[cout][k] . P[k][pixel]: weights are the
// A operand, a tile row of 32 pixels the B operand of v_mfma_f32_32x32x16_{bf16,f16} / 4 x v_mfma_f32_32x32x2_f32, so a lane owns
// one pixel and 4 consecutive couts per register quad), GroupNorm + ReLU (+ condition + E[t] / DDIM update) of the PREVIOUS layer
// applied while the halo'd input patch is staged into LDS, bias + this layer's GroupNorm partial sums in the epilogue.
// Structured as a software pipeline:
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
//   * LDS rows are unpadded (32 / 64 / 128 B) with a 16-B-piece XOR swizzle keyed on the patch COLUMN
//     (weights: the cout row):  piece' = piece ^ ((k / R) & (PPP-1)),  R = 256 / ROWB,  PPP = ROWB / 16.
//     A 32-pixel MFMA block is one tile row, lane == column, so every 16-lane ds_read_b128 group hits 16
//     distinct 16-B slots of the 256-B bank row (k mod 16 <-> slot is a bijection) and the whole LDS
//     address is (per-lane column term, 3 variants of dx) + (wave-uniform / immediate row term): no
//     per-tap VALU address math.  conv2 / conv3 stay <= 77 KiB of LDS (two workgroups per CU).
//   * the prologue is VALU-lean (rocprof PMC showed the first cut VALU-bound: 7.4k VALU vs 576 MFMA per
//     wave in conv3): staging geometry is computed once per kernel, each thread keeps its slice of the
//     GroupNorm table in registers, packing uses v_cvt_pk_{bf16,f16}_f32.
//   * bf16 / f16 outputs are stored 16 B per lane: `v_permlane32_swap` pairs the two half-waves'
//     4-cout quads into 8 consecutive couts (cdna guide T21).
#include "dd_igemm2_cfg.h"
#include "dd_gcn.h"

namespace dd {

#if DD_PHASE_PROF && !defined(DD_HOST_EMULATION)
#define DD_PROF_MARK(i) do { if (p.prof != nullptr && threadIdx.x == 0) p.prof[(size_t)blockIdx.x * 8 + (i)] = (unsigned long long)wall_clock64(); } while (0)
#define DD_PROF_HWID() do { if (p.prof != nullptr && threadIdx.x == 0) {                                                     \
    unsigned hw_, xcc_; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_)); asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));   \
    p.prof[(size_t)blockIdx.x * 8 + 6] = hw_; p.prof[(size_t)blockIdx.x * 8 + 7] = xcc_; } } while (0)
// per-wave cycle accounting inside the main loop (shader clock): [0] MFMA blocks (fragment reads + MFMA issue), [1] next-chunk prologue
// transform, [2] waits + barrier, [3] DMA / raw-load issue; stored behind the per-workgroup records: prof[gridDim.x * 8 + (wg * WAVES + wave) * 4 + k]
#define DD_PROF_CLK() ((unsigned long long)__builtin_amdgcn_s_memtime())
#define DD_PROF_LOOP_DECL unsigned long long pl_acc_[4] = {0, 0, 0, 0}, pl_t_ = 0
#define DD_PROF_LOOP_START() do { if (p.prof != nullptr) pl_t_ = DD_PROF_CLK(); } while (0)
#define DD_PROF_LOOP_ADD(k) do { if (p.prof != nullptr) { const unsigned long long n_ = DD_PROF_CLK(); pl_acc_[k] += n_ - pl_t_; pl_t_ = n_; } } while (0)
#define DD_PROF_LOOP_STORE() do { if (p.prof != nullptr && lane == 0) { for (int k_ = 0; k_ < 4; ++k_)                          \
    p.prof[(size_t)gridDim.x * 8 + ((size_t)blockIdx.x * C::WAVES + wave) * 4 + k_] = pl_acc_[k_]; } } while (0)
#else
#define DD_PROF_MARK(i) ((void)0)
#define DD_PROF_HWID() ((void)0)
#define DD_PROF_LOOP_DECL
#define DD_PROF_LOOP_START() ((void)0)
#define DD_PROF_LOOP_ADD(k) ((void)0)
#define DD_PROF_LOOP_STORE() ((void)0)
#endif

template <class C>
__global__ void __launch_bounds__(C::THREADS, C::MIN_WAVES_PER_SIMD) conv_igemm2_kernel(ConvParams p) {
  constexpr int EK = C::EK;
  constexpr int PW = C::PW, ROWB = C::ROWB, PPP = C::PPP, RPB = C::RPB, EPP = C::EPP, CK = C::CK;
  constexpr int NIT = C::NIT, NLD = C::NLD, IN_ESZ = C::IN_ESZ, NKQ = C::NKQ;
  DD_DYN_SMEM(smem);
  float* s_tab = reinterpret_cast<float*>(smem + C::NPB * C::PATCH_BYTES + C::NWB * C::W_BYTES);
  float* tab_a = s_tab;
  float* tab_b = s_tab + C::CTAB;
  float* tab_e = s_tab + 2 * C::CTAB;                 // PRO_GN_ADD only (NTABS == 3)
  float* tab_bias = s_tab + C::NTABS * C::CTAB;
  float* tab_et = tab_bias + C::NT * C::SPW;  // [10][64] (ADD_C only)
  double* s_red = reinterpret_cast<double*>(tab_et + (C::ADD_C ? 10 * HID_C : 0));   // [WAVES][8] statistics scratch (16-B aligned: every term is)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, g = lane >> 5;
  DD_PROF_MARK(0);
  DD_PROF_HWID();

  // XCD-aware workgroup -> tile map (cdna guide T1): the dispatcher places block b on XCD b % 8, each XCD has its own
  // L2.  Remap so that every XCD owns one contiguous run of (tile, cout-split) work items: the cout halves of a tile
  // and vertically / horizontally adjacent tiles (which share halo rows) then hit the same L2 close in time.
  constexpr int NSPLIT = C::COUT_PAD / C::NT;
  int wgid;
  {
    const int nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);     // bijective for any nwg
  }
  // one workgroup per (tile, SPW consecutive cout splits)
  constexpr int NSPLIT_G = NSPLIT / C::SPW;
  const int nsplit = (wgid % NSPLIT_G) * C::SPW;      // first split of this workgroup
  const int tiles_per_img = p.tiles_x * p.tiles_y;
#ifndef DD_SPLIT_FLIP
#define DD_SPLIT_FLIP 1
#endif
  // DD_SPLIT_FLIP (round 6): workgroups of odd tiles walk their two cout splits in the opposite order, so that at any moment half the workgroups stream the first
  // split's weight stages and half the second's (otherwise every workgroup of an XCD reads the same L2 lines at the same time).  Bit-identical outputs; conv2 130.7 /
  // 131.4 -> 129.4 / 129.0 us at KITTI B = 4 (A/B/A/B on one box, profiles/r06_call19_split_flip.txt): -1 %.
  const int flip = (DD_SPLIT_FLIP && C::SPW == 2) ? ((wgid / NSPLIT_G) & 1) : 0;
  int n0 = (nsplit + flip) * C::NT;                   // first cout of the split being computed
  int sbase = 0;                                      // weight stages consumed by the splits already done (ring slots run on across splits)
  const int h = p.h, w = p.w;
  const bool have_norm = (C::PRO == PRO_X) ? (p.step > 0) : (C::PRO != PRO_RAW);
  const int abl = DD_ABLATE ? p.ablate : 0;   // timing experiments are compiled in with -DDD_ABLATE=1 only
  if (abl & 256) return;                  // timing floor: launch + dispatch only
  int tile = wgid / NSPLIT_G;
  int b, y0, x0;
  const char *in_b, *cond_b, *y4_b;
  char* xout_b;
  auto set_tile_base = [&](int t) {
    tile = t;
    b = t / tiles_per_img;
    const int trem = t - b * tiles_per_img;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    y0 = ty * C::TH; x0 = tx * C::TW;
    if constexpr (C::IS_NECK) {
      const int cs = p.in_cstride ? p.in_cstride : C::CIN;
      in_b = reinterpret_cast<const char*>(p.in) + ((size_t)b * cs + p.in_coff) * h * w * IN_ESZ;
    } else {
      in_b = reinterpret_cast<const char*>(p.in) + (size_t)b * h * w * C::CIN * IN_ESZ;
    }
    cond_b = (C::PRO == PRO_GN_ADD) ? reinterpret_cast<const char*>(p.cond) + (size_t)b * h * w * C::CIN * IN_ESZ : nullptr;
    y4_b = (C::PRO == PRO_X) ? reinterpret_cast<const char*>(p.y4) + (size_t)b * h * w * LATENT_C * 4 : nullptr;
    xout_b = (C::PRO == PRO_X) ? reinterpret_cast<char*>(p.xout) + (size_t)b * h * w * LATENT_C * 4 : nullptr;
  };
  set_tile_base(tile);

  // ---- weight stage s -> ring slot (s & 1) by LDS-DMA: wave `wave` copies KiB-chunks wave, wave+WAVES, ... ----
  // Issued through inline asm (M0 = wave-uniform LDS destination, lane l lands at M0 + 16*l): with the
  // __builtin form hipcc treats the DMA as an LDS write that may alias every later ds_read and drains
  // `s_waitcnt vmcnt(0)` right after issuing it, which serialises DMA latency with the MFMAs.  hipcc does not count
  // asm VMEM ops, so every wait for the DMA below is explicit; VMEM loads retire in issue order, so hipcc's own
  // counted waits for the raw patch loads only become more conservative (cdna guide 5.7).
  const unsigned lds_base = DD_LDS_BASE(smem);
  constexpr int NWPIECE = (C::W_BYTES / 1024 + C::WAVES - 1) / C::WAVES;      // 1-KiB DMA instructions per wave and stage
  auto issue_weight_piece = [&](int s, int c) {
    const int si = flip ? (s < C::NSTAGE ? s + C::NSTAGE : s - C::NSTAGE) : s;      // (DD_SPLIT_FLIP: the stage's place in the packed image)
    const char* src = reinterpret_cast<const char*>(p.wpack) + ((size_t)nsplit * C::NSTAGE + si) * (size_t)C::W_BYTES + lane * 16;
    const unsigned dst = lds_base + C::W_OFF + (s & (C::NWB - 1)) * C::W_BYTES;
    const int kc = c * C::WAVES + wave;
    if (kc < C::W_BYTES / 1024) {
      const char* gsrc = src + (size_t)kc * 1024;
      const unsigned ldst = __builtin_amdgcn_readfirstlane(dst + kc * 1024);
      DD_LDS_DMA16(smem, gsrc, ldst);
    }
  };
  auto issue_weights = [&](int s) {
#pragma unroll
    for (int c = 0; c < NWPIECE; ++c) issue_weight_piece(s, c);
  };

  // ---- per-thread staging geometry: independent of the channel chunk, computed once ---------------------
  // item u of this thread = 16-B piece `jfix` of patch pixel pp(u); THREADS % PPP == 0 keeps jfix constant
  const int jfix = tid & (PPP - 1);
  int lds_off[NIT];            // byte offset of the (swizzled) piece inside a patch buffer
  int pix_off[NIT];            // clamped global pixel index gy*w+gx
  unsigned m_valid = 0, m_inside = 0, m_interior = 0;
  auto set_tile_geometry = [&]() {
    m_valid = 0; m_inside = 0; m_interior = 0;
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      const int it = u * C::THREADS + tid;
      const int itc = it < C::ITEMS ? it : C::ITEMS - 1;
      const int pp = itc >> C::LOG2_PPP;
      const int pr = pp / PW, pc = pp - pr * PW;
      const int gy = y0 - C::HALO + pr, gx = x0 - C::HALO + pc;
      const bool inside = gy >= 0 && gy < h && gx >= 0 && gx < w;
      if (it < C::ITEMS) m_valid |= 1u << u;
      if (inside) m_inside |= 1u << u;
      if (inside && pr >= C::HALO && pr < C::HALO + C::TH && pc >= C::HALO && pc < C::HALO + C::TW) m_interior |= 1u << u;
      const int gyc = gy < 0 ? 0 : (gy >= h ? h - 1 : gy);
      const int gxc = gx < 0 ? 0 : (gx >= w ? w - 1 : gx);
      pix_off[u] = gyc * w + gxc;
      lds_off[u] = pp * ROWB + ((jfix << 4) ^ swz16<RPB, PPP>(pc));
    }
  };
  set_tile_geometry();

  // ---- raw patch fetch of one channel chunk into registers (unconditional, clamped addresses) --------
  constexpr int RD = C::RAW_DEPTH;     // raw-patch register slots: chunk c lives in slot c % RD (RD = 2: fetched two chunks ahead)
  uint4 raw[RD][NIT][NLD];
  uint4 aux[NIT][NLD];
  auto load_raw = [&](int chunk, int slot) {
    const int cbase = chunk * CK + jfix * EPP;
    const size_t off0 = (C::CIN >= ACT_CB) ? ((size_t)(cbase >> 5) * h * w * ACT_CB + (cbase & (ACT_CB - 1))) : (size_t)cbase;
    if constexpr (C::IN_NCHW) {
      // the caller's NCHW fp32 tensor: channel c of this image is the plane at c * h * w; the piece's eight channels are eight 4-byte loads
      // (lanes 0, 2, 4, ... = consecutive pixels of plane cbase + i, the odd lanes of plane cbase + 8 + i: two runs of 128 contiguous bytes)
      static_assert(EPP == 8 && NLD == 2 && C::PRO == PRO_RAW, "fp32 source, eight channels per piece, no auxiliary tensor");
      const size_t HW = (size_t)h * w;
#pragma unroll
      for (int u = 0; u < NIT; ++u) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(in_b) + (size_t)cbase * HW + pix_off[u];
#pragma unroll
        for (int q = 0; q < NLD; ++q)
          raw[slot][u][q] = make_uint4(src[(4 * q + 0) * HW], src[(4 * q + 1) * HW], src[(4 * q + 2) * HW], src[(4 * q + 3) * HW]);
      }
      DD_VMEM_LOADS_ISSUED(NIT * NLD * 4);
      return;
    }
#pragma unroll
    for (int u = 0; u < NIT; ++u) {
      const size_t goff = (off0 + (size_t)pix_off[u] * C::PIXSTRIDE) * IN_ESZ;
#pragma unroll
      for (int q = 0; q < NLD; ++q) raw[slot][u][q] = *reinterpret_cast<const uint4*>(in_b + goff + q * 16);
      if constexpr (C::PRO == PRO_GN_ADD) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) aux[u][q] = *reinterpret_cast<const uint4*>(cond_b + goff + q * 16);
      } else if constexpr (C::PRO == PRO_X) {
        if (have_norm) {
#pragma unroll
          for (int q = 0; q < NLD; ++q) aux[u][q] = *reinterpret_cast<const uint4*>(y4_b + goff + q * 16);
        }
      }
    }
    DD_VMEM_LOADS_ISSUED(NIT * NLD * ((C::PRO == PRO_GN_ADD || (C::PRO == PRO_X && have_norm)) ? 2 : 1));   // host model only (dd_gcn.h)
  };

  float c1 = 1.f, c2 = 0.f;
  // ---- registers -> normalise -> swizzled LDS patch buffer at byte offset pbuf_off ---------------------
  // One staging item (16-B piece `jfix` of patch pixel u).  This thread only ever touches channels [c0, c0+EPP), so its
  // slice of the GroupNorm table is (re)read from LDS as three/two float4 pairs.
  // This thread's slice of the GroupNorm table for one channel chunk: read from LDS ONCE per chunk (every item of the chunk touches the
  // same EPP channels).  Inside the item loop the compiler cannot keep it: the patch stores alias the table through `smem`.
  float ta[EPP], tb[EPP], te[EPP];
  auto load_table_slice = [&](int chunk) {
    if constexpr (C::PRO == PRO_RAW) return;
    const int c0 = (C::PRO == PRO_X) ? jfix * EPP : chunk * CK + jfix * EPP;
    if (have_norm) {
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
  };
  auto transform_item = [&](int chunk, int pbuf_off, int u, int slot) {
    if (!((m_valid >> u) & 1u)) return;
    if constexpr (C::PRO == PRO_RAW && !C::SPLIT) {
      // no normalisation between the producer and this convolution: the stored elements are the operands
      *reinterpret_cast<uint4*>(smem + pbuf_off + lds_off[u]) = ((m_inside >> u) & 1u) ? raw[slot][u][0] : make_uint4(0u, 0u, 0u, 0u);
      return;
    }
    const int c0 = (C::PRO == PRO_X) ? jfix * EPP : chunk * CK + jfix * EPP;
    float v[EPP];
    if ((m_inside >> u) & 1u) {
      if constexpr (C::PRO == PRO_X) {
        // DDIM update of the previous step fused into the load: x <- c1*x + c2*relu(gn4(y4))
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
          const uint32_t xw[4] = {raw[slot][u][q].x, raw[slot][u][q].y, raw[slot][u][q].z, raw[slot][u][q].w};
          const uint32_t yw[4] = {aux[u][q].x, aux[u][q].y, aux[u][q].z, aux[u][q].w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float x = __builtin_bit_cast(float, xw[i]);
            if (have_norm) {
              const float e = fmaxf(fmaf(ta[q * 4 + i], __builtin_bit_cast(float, yw[i]), tb[q * 4 + i]), 0.f);
              x = c1 * x + c2 * e;
            }
            v[q * 4 + i] = x;
          }
        }
        if (have_norm && ((m_interior >> u) & 1u)) {
          const size_t goff = ((size_t)pix_off[u] * LATENT_C + c0) * 4;
#pragma unroll
          for (int q = 0; q < NLD; ++q)
            *reinterpret_cast<float4*>(xout_b + goff + q * 16) = make_float4(v[q * 4], v[q * 4 + 1], v[q * 4 + 2], v[q * 4 + 3]);
        }
      } else if constexpr (C::SPLIT) {
        // fp32 tensors in HBM: the piece's 8 channels are two 16-byte loads; the GroupNorm table already carries SPLIT_PSCALE, raw inputs
        // (no bound on their range) stay unscaled
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
          const uint32_t yw[4] = {raw[slot][u][q].x, raw[slot][u][q].y, raw[slot][u][q].z, raw[slot][u][q].w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float y = __builtin_bit_cast(float, yw[i]);
            if constexpr (C::PRO == PRO_RAW) v[q * 4 + i] = y;
            else v[q * 4 + i] = fmaxf(fmaf(ta[q * 4 + i], y, tb[q * 4 + i]), 0.f);
          }
          if constexpr (C::PRO == PRO_GN_ADD) {
            const uint32_t cw[4] = {aux[u][q].x, aux[u][q].y, aux[u][q].z, aux[u][q].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) v[q * 4 + i] = fmaf(__builtin_bit_cast(float, cw[i]), SPLIT_PSCALE, v[q * 4 + i] + te[q * 4 + i]);
          }
        }
      } else {
        if constexpr (C::PRO == PRO_GN && EK != EK_F32) {
          *reinterpret_cast<uint4*>(smem + pbuf_off + lds_off[u]) = affine_relu_pack<C::IN_K, EK>(raw[slot][u][0], ta, tb);
          return;
        }
        Piece<C::IN_K>::unpack(raw[slot][u][0], v);
#pragma unroll
        for (int i = 0; i < EPP; ++i) v[i] = fmaxf(fmaf(ta[i], v[i], tb[i]), 0.f);
        if constexpr (C::PRO == PRO_GN_ADD) {
          float cv[EPP];
          Piece<C::IN_K>::unpack(aux[u][0], cv);
#pragma unroll
          for (int i = 0; i < EPP; ++i) v[i] = v[i] + (cv[i] + te[i]);
        }
      }
    } else {
#pragma unroll
      for (int i = 0; i < EPP; ++i) v[i] = 0.f;       // zero padding applies AFTER the normalisation
    }
    if constexpr (C::NPLP == 2) {
      // operand pair: hi = f16(v), lo = f16(v - hi); the two planes of the patch buffer
      const uint4 hi = Piece<EK_F16>::pack(v);
      float vh[EPP];
      Piece<EK_F16>::unpack(hi, vh);
#pragma unroll
      for (int i = 0; i < EPP; ++i) vh[i] = v[i] - vh[i];
      *reinterpret_cast<uint4*>(smem + pbuf_off + lds_off[u]) = hi;
      *reinterpret_cast<uint4*>(smem + pbuf_off + C::PATCH_PLANE + lds_off[u]) = Piece<EK_F16>::pack(vh);
    } else {
      *reinterpret_cast<uint4*>(smem + pbuf_off + lds_off[u]) = Piece<EK>::pack(v);
    }
  };
  auto transform_write = [&](int chunk, int pbuf_off, int slot) {
    load_table_slice(chunk);
#pragma unroll
    for (int u = 0; u < NIT; ++u) transform_item(chunk, pbuf_off, u, slot);
  };

  // ---- kick off every independent load at once: weights of stage 0 (LDS-DMA), the GroupNorm partial sums
  //      of the producing layer, this thread's gamma/beta/embedding entries, the raw patch of chunk 0 ------
  // Loads only, in the order they are needed, and nothing that waits in between (VMEM returns in issue order: a wait for a late small
  // load is a wait for every large one in front of it): small table inputs, then the patch, then the accumulators' start values.
  issue_weights(0);
  float my_bias = p.bias[nsplit * C::NT + (tid < C::NT * C::SPW ? tid : 0)];     // into tab_bias behind the loads below (the table holds the splits in their natural order)
  if constexpr (C::ADD_T) my_bias += p.ttab[(size_t)b * p.ttab_bstride + (tid < HID_C ? tid : 0)];        // hoisted Swin form: + the E[t] term of the reference border class
  // hoisted condition term: this thread's entries of the E[t] tap-sum row.  Only the LOADS are issued here (into registers): the LDS
  // image is written behind the GroupNorm butterfly and read in the epilogue, so the timestep -> etab row -> LDS dependency does
  // not sit in front of the partial-sum / patch / accumulator loads (it cost ~4 us of every workgroup: profiles/history/r02_run3_phase_profile.md)
  constexpr int NET = C::ADD_C ? (10 * HID_C + C::THREADS - 1) / C::THREADS : 1;
  float et_r[NET];
  if constexpr (C::ADD_C) {
    const long long t = (DD_T_KNOWN && p.t_known >= 0) ? p.t_known : clamp_t(p.tvec[p.t_base + b * p.t_bstride]);
#pragma unroll
    for (int k = 0; k < NET; ++k) {
      const int i = k * C::THREADS + tid;
      et_r[k] = (i < 10 * HID_C) ? p.etab[(size_t)t * 10 * HID_C + i] : 0.f;
    }
  }
  double2 sv0 = make_double2(0.0, 0.0), sv1 = make_double2(0.0, 0.0);
  float my_gamma = 0.f, my_beta = 0.f, my_emb = 0.f;
  auto load_norm_inputs = [&]() {          // everything the GroupNorm table of image b is built from
    if (!have_norm) return;
    // 32 slots x 4 groups x (sum, sumsq): lane l takes slot l>>1, groups 2*(l&1) and 2*(l&1)+1 (every wave redundantly)
    const double* st = p.stats_in + (size_t)b * STAT_SLOTS * STAT_STRIDE + (lane >> 1) * STAT_STRIDE + (lane & 1) * 4;
    sv0 = *reinterpret_cast<const double2*>(st);
    sv1 = *reinterpret_cast<const double2*>(st + 2);
    {
      // unconditional (clamped) so that no branch ties a wait to the load: threads past CTAB never use what they read
      const int ct = (C::CTAB > 0 && tid < C::CTAB) ? tid : 0;
      my_gamma = p.gn_gamma[ct];
      my_beta = p.gn_beta[ct];
      if constexpr (C::PRO == PRO_GN_ADD) {
        const long long t = (DD_T_KNOWN && p.t_known >= 0) ? p.t_known : clamp_t(p.tvec[p.t_base + b * p.t_bstride]);
        my_emb = p.emb[(size_t)t * COND_C + ct];
      }
    }
    if constexpr (C::PRO == PRO_X) { c1 = p.c1c2[2 * (p.step - 1)]; c2 = p.c1c2[2 * (p.step - 1) + 1]; }
  };
  load_norm_inputs();
  DD_SCHED_FENCE();                         // table inputs first: the butterfly below waits for them and for nothing issued after them
  load_raw(0, 0);
  if constexpr (RD == 2 && C::NCHUNK > 1) load_raw(1, 1);
  DD_SCHED_FENCE();                         // the patch in front of the accumulators' start values

  // per-lane LDS addressing.  Pixel block (wave*WM + m) is tile row (wave*WM + m), lane li is tile column li,
  // so tap (dy,dx) reads patch row (wave*WM + m + dy), patch column (li + dx): everything row-dependent is an
  // immediate or a wave-uniform scalar; only the column term (3 variants of dx) lives in VGPRs.
  const int g16 = g << 4;
  constexpr int NDX = C::KS == 5 ? 5 : 3;
  int colt[NDX][NKQ];
#pragma unroll
  for (int dx = 0; dx < NDX; ++dx)
#pragma unroll
    for (int kq = 0; kq < NKQ; ++kq)
      colt[dx][kq] = wave * (C::WM * PW * ROWB) + (li + dx) * ROWB + (((kq << 5) | g16) ^ swz16<RPB, PPP>(li + dx));
  int wkt[NKQ];
#pragma unroll
  for (int kq = 0; kq < NKQ; ++kq) wkt[kq] = C::W_OFF + li * ROWB + (((kq << 5) | g16) ^ swz16<RPB, PPP>(li));

  const int e_b = b, e_y0 = y0, e_x0 = x0, e_tile = tile;     // the tile the accumulators / epilogue belong to
  f32x16_t acc[C::WN][C::WM];
  // accumulators.  With the hoisted condition term they START at conv3(cond)[pixel][cout] (fp32, D-fragment order): the
  // loads fly with everything else above and need no extra registers or epilogue traffic.
  auto init_acc = [&]() {
#pragma unroll
  for (int m = 0; m < C::WM; ++m) {
    const int gy = y0 + wave * C::WM + m, gx = x0 + li;
    const bool pv = gy < h && gx < w;
#pragma unroll
    for (int n = 0; n < C::WN; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 cv = make_float4(0.f, 0.f, 0.f, 0.f);
        if constexpr (C::ADD_ACC) {
          // layer 8 stored conv3(cond) in accumulator-fragment order: every load is one contiguous KiB (f16: half a KiB) per wave
          (void)pv;
          const size_t fi = ((((size_t)tile * C::WAVES + wave) * C::WN + n) * C::WM + m) * 256 + q * 64 + lane;
          if constexpr (C::Q15) {
            // int16 quads, one fp32 scale per (tile, wave, n, m) block of 32 pixels x 32 couts: a wave-uniform scalar
            const uint2 h2 = reinterpret_cast<const uint2*>(p.cadd)[fi];
            const float sc = p.cadd_scale[(((size_t)tile * C::WAVES + wave) * C::WN + n) * C::WM + m];
            cv = make_float4((float)(short)(h2.x & 0xFFFFu) * sc, (float)((int)h2.x >> 16) * sc, (float)(short)(h2.y & 0xFFFFu) * sc, (float)((int)h2.y >> 16) * sc);
          } else if constexpr (C::CADD16) {
            const uint2 h2 = reinterpret_cast<const uint2*>(p.cadd)[fi];
            cv = make_float4(f16_to_f32(h2.x & 0xFFFFu), f16_to_f32(h2.x >> 16), f16_to_f32(h2.y & 0xFFFFu), f16_to_f32(h2.y >> 16));
          } else {
            cv = reinterpret_cast<const float4*>(p.cadd)[fi];
          }
        }
        if constexpr (C::SPLIT && C::ADD_ACC) { constexpr float IS = 1.f / split_oscale(C::PRO); cv.x *= IS; cv.y *= IS; cv.z *= IS; cv.w *= IS; }   // accumulators run at the operands' scale
        acc[n][m][q * 4 + 0] = cv.x; acc[n][m][q * 4 + 1] = cv.y; acc[n][m][q * 4 + 2] = cv.z; acc[n][m][q * 4 + 3] = cv.w;
      }
  }
  };
  init_acc();
  DD_SCHED_FENCE();                         // every load above is in flight before the first wait below
  if (tid < C::NT * C::SPW) tab_bias[tid] = my_bias;     // visible after the first barrier below
  if (abl & 512) { DD_WAIT_VM(0); if (raw[0][0][0].x == 0x12345678u && sv0.x == 1.5) p.xout[0] = my_gamma; return; }

  // ---- GroupNorm affine table: butterfly over the 32 slots inside each wave, then one channel per thread ----
  if (have_norm) {
#pragma unroll
    for (int off = 2; off <= 32; off <<= 1) {
      sv0.x += __shfl_xor(sv0.x, off, 64); sv0.y += __shfl_xor(sv0.y, off, 64);
      sv1.x += __shfl_xor(sv1.x, off, 64); sv1.y += __shfl_xor(sv1.y, off, 64);
    }
    const double2 ov0 = make_double2(__shfl_xor(sv0.x, 1, 64), __shfl_xor(sv0.y, 1, 64));
    const double2 ov1 = make_double2(__shfl_xor(sv1.x, 1, 64), __shfl_xor(sv1.y, 1, 64));
    const bool hi = lane & 1;                  // this lane reduced groups 2,3 (hi) or 0,1
    const double2 g0 = hi ? ov0 : sv0, g1 = hi ? ov1 : sv1, g2 = hi ? sv0 : ov0, g3 = hi ? sv1 : ov1;
    // Non-finite statistics (a NaN / Inf anywhere in the producing layer's output of this image -- e.g. a NaN in the caller's condition map): the
    // reference's GroupNorm then returns NaN for the whole image and its ReLU keeps it (torch.relu(NaN) = NaN).  Here the ReLU is v_max_f32, which
    // DROPS a NaN operand, so without help the image would come out finite and wrong.  Every thread holds all four groups' sums (the butterfly is
    // redundant per wave), so each decides locally: poisoned -> this layer's bias (conv1: also c2 of the DDIM update) becomes NaN, every output of
    // the image and its statistics are NaN, and the next layer does the same -- down to the final kernel, which writes the NaN image (dd_misc.hip).
    const bool poisoned = !(((g0.x - g0.x) + (g0.y - g0.y)) + ((g1.x - g1.x) + (g1.y - g1.y)) + ((g2.x - g2.x) + (g2.y - g2.y)) + ((g3.x - g3.x) + (g3.y - g3.y)) == 0.0);
    if (poisoned) {
      if (tid < C::NT * C::SPW) tab_bias[tid] = __builtin_nanf("");      // (program-ordered behind this thread's own bias store above; read after the barrier below)
      if constexpr (C::PRO == PRO_X) c2 = __builtin_nanf("");
    }
    static_assert(C::PRO == PRO_RAW || C::CTAB <= C::THREADS, "one table channel per thread");
    if (tid < C::CTAB) {
      constexpr int CG = (C::CTAB > 0) ? C::CTAB / GN_GROUPS : 1;
      const int grp = tid / CG;
      const double2 gs = grp == 0 ? g0 : grp == 1 ? g1 : grp == 2 ? g2 : g3;
      const double inv_cnt = 1.0 / ((double)h * (double)w * (double)CG);
      const double mean = gs.x * inv_cnt;
      double var = gs.y * inv_cnt - mean * mean;                 // biased, as torch
      var = var > 0.0 ? var : 0.0;
      const double a = (double)my_gamma / sqrt(var + (double)GN_EPS);
      // split f16: behind a GroupNorm the patch is carried times SPLIT_PSCALE (exact power of two; relu commutes with it); conv1's table
      // produces the state that is written back and its input has no bound: unscaled (dd_kernels.h)
      constexpr float TS = C::SPLIT ? split_pscale(C::PRO) : 1.f;
      tab_a[tid] = (float)a * TS;
      tab_b[tid] = (float)((double)my_beta - mean * a) * TS;
      if constexpr (C::PRO == PRO_GN_ADD) tab_e[tid] = my_emb * TS;
    }
    if constexpr (C::ADD_C) {
#pragma unroll
      for (int k = 0; k < NET; ++k) {
        const int i = k * C::THREADS + tid;
        if (i < 10 * HID_C) tab_et[i] = et_r[k];
      }
    }
    __syncthreads();                       // table visible
    if constexpr (C::ADD_C) {
      if (tid < HID_C) tab_bias[tid] += tab_et[9 * HID_C + tid];     // read again only in the epilogue (many barriers later)
    }
  }
  DD_PROF_MARK(1);
  transform_write(0, 0, 0);
  DD_WAIT_VM(0);   // this wave's DMA pieces of weight stage 0 have landed
  __syncthreads();                                    // patch 0 and weight stage 0 are in LDS for everybody
  DD_PROF_MARK(2);
  if (abl & 1024) return;
  // ---- the MFMAs of stage (chunk, tg): TG taps x NKQ k-steps x (WM x WN) tiles out of LDS ------------------
  auto mfma_block = [&](int chunk, int tg) {
#if DD_SETPRIO && !defined(DD_HOST_EMULATION)
    __builtin_amdgcn_s_setprio(DD_SETPRIO);      // the wave inside its MFMA block wins issue arbitration against the co-resident workgroup's staging wave
#endif
    const int s = sbase + chunk * C::NTG + tg;
    const int poff = (chunk & (C::NPB - 1)) * C::PATCH_BYTES;
    const int woff = (s & (C::NWB - 1)) * C::W_BYTES;
    int wa[NKQ];
#pragma unroll
    for (int kq = 0; kq < NKQ; ++kq) wa[kq] = wkt[kq] + woff;
    {
      // Software-pipelined fragment stream: the ds_reads of group g+1 (one (tap, k-step) = WM pixel + WN weight fragments)
      // are issued between the MFMAs of group g, into the other half of a two-deep register buffer, so a wave's MFMA
      // stream does not wait for LDS latency inside a stage (only the first group of a stage is exposed).
      constexpr int NG = C::TG * NKQ;          // fragment groups per stage
      constexpr int NF = C::NPLP * C::WM + C::NPL * C::WN;        // fragments (ds_read_b128) per group; split f16: [pixel hi | pixel lo | weight hi | weight lo]
      constexpr int NM = C::WM * C::WN * ((EK == EK_F32) ? 4 : C::SPLIT ? (C::WONLY ? 2 : 3) : 1);   // MFMA instructions per group
      constexpr int FD = C::FRAG_DEPTH;        // register buffers: FD-1 groups of ds_reads are in flight ahead of the MFMAs
      uint4 fr[FD][NF];
      auto load_group = [&](int gi, uint4 (&f)[NF]) {
        const int t = gi / NKQ, kq = gi % NKQ;
        const int dy = (C::KS == 1) ? 0 : (C::TG == 9) ? t / 3 : (C::TG == 3 || C::TG == 5) ? tg : tg / 3;      // (5x5: one kernel row per stage)
        const int dx = (C::KS == 1) ? 0 : (C::TG == 9) ? t % 3 : (C::TG == 3 || C::TG == 5) ? t : tg % 3;
        const int pa = colt[dx][kq] + poff + dy * (PW * ROWB);
#pragma unroll
        for (int m = 0; m < C::WM; ++m) f[m] = *reinterpret_cast<const uint4*>(smem + pa + m * (PW * ROWB));
        if constexpr (C::NPLP == 2) {
#pragma unroll
          for (int m = 0; m < C::WM; ++m) f[C::WM + m] = *reinterpret_cast<const uint4*>(smem + pa + C::PATCH_PLANE + m * (PW * ROWB));
        }
#pragma unroll
        for (int n = 0; n < C::WN; ++n) f[C::NPLP * C::WM + n] = *reinterpret_cast<const uint4*>(smem + wa[kq] + (t * C::NT + n * 32) * ROWB);
        if constexpr (C::SPLIT) {
#pragma unroll
          for (int n = 0; n < C::WN; ++n) f[C::NPLP * C::WM + C::WN + n] = *reinterpret_cast<const uint4*>(smem + wa[kq] + C::W_PLANE + (t * C::NT + n * 32) * ROWB);
        }
      };
      if constexpr (FD > 1) {
#pragma unroll
        for (int pg = 0; pg < FD - 1 && pg < NG; ++pg) load_group(pg, fr[pg % FD]);
        __builtin_amdgcn_sched_group_barrier(0x100, NF * ((FD - 1 < NG) ? FD - 1 : NG), 0);
      }
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        const bool pre = FD > 1 && gi + FD - 1 < NG;
        if constexpr (FD > 1) {
          if (pre) load_group(gi + FD - 1, fr[(gi + FD - 1) % FD]);
        } else {
          load_group(gi, fr[0]);
        }
        if constexpr (C::SPLIT) {
          // W.P = Whi.Phi + Whi.Plo + Wlo.Phi (the product kind outermost: consecutive MFMAs hit different accumulators)
          constexpr int PH_ = 0, PL_ = C::WM, WH_ = C::NPLP * C::WM, WL_ = C::NPLP * C::WM + C::WN;
#pragma unroll
          for (int n = 0; n < C::WN; ++n)
#pragma unroll
            for (int m = 0; m < C::WM; ++m) mma_step<EK>(acc[n][m], fr[gi % FD][WH_ + n], fr[gi % FD][PH_ + m]);
          if constexpr (!C::WONLY) {      // (weights-only pair: the patch has one plane)
#pragma unroll
          for (int n = 0; n < C::WN; ++n)
#pragma unroll
            for (int m = 0; m < C::WM; ++m) mma_step<EK>(acc[n][m], fr[gi % FD][WH_ + n], fr[gi % FD][PL_ + m]);
          }
#pragma unroll
          for (int n = 0; n < C::WN; ++n)
#pragma unroll
            for (int m = 0; m < C::WM; ++m) mma_step<EK>(acc[n][m], fr[gi % FD][WL_ + n], fr[gi % FD][PH_ + m]);
        } else {
#pragma unroll
        for (int n = 0; n < C::WN; ++n)
#pragma unroll
          for (int m = 0; m < C::WM; ++m) mma_step<EK>(acc[n][m], fr[gi % FD][C::WM + n], fr[gi % FD][m]);
        }
        if constexpr (FD == 1) {
        } else if (pre) {
          // one ds_read behind each of the first NF MFMAs, the rest of the MFMAs after them
          constexpr int PER = (NM >= NF) ? 1 : (NF + NM - 1) / NM;
#pragma unroll
          for (int i = 0; i < ((NM >= NF) ? NF : NM); ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, PER, 0);
          }
          if (NM > NF) __builtin_amdgcn_sched_group_barrier(0x008, NM - NF, 0);
        } else {
          __builtin_amdgcn_sched_group_barrier(0x008, NM, 0);
        }
      }
    }
#if DD_SETPRIO && !defined(DD_HOST_EMULATION)
    __builtin_amdgcn_s_setprio(0);
#endif
  };

  constexpr int NRAW = NIT * NLD * (C::IN_NCHW ? 4 : (C::PRO == PRO_GN || C::PRO == PRO_RAW) ? 1 : 2);   // raw-patch loads per load_raw()
  static_assert(NRAW <= 63, "vmcnt field");

  // par = chunk % RD as a compile-time constant at every call site (static register indexing of the raw slots)
  DD_PROF_LOOP_DECL;
  DD_PROF_LOOP_START();
  auto stage = [&](int chunk, int tg, int par) {
    const int s = sbase + chunk * C::NTG + tg;
    const bool want_w = s + 1 < C::NSTAGE * C::SPW && !(abl & 4);
    const bool want_raw = C::NCHUNK > 1 && tg == 0 && chunk + RD < C::NCHUNK && !(abl & 2);
    if (want_w) issue_weights(s + 1);
    asm volatile("" ::: "memory");         // keep the DMA ahead of the raw loads in issue order (counted vmcnt below)
    if (want_raw) load_raw(chunk + RD, par);
    DD_PROF_LOOP_ADD(3);
    if (!(abl & 8)) mfma_block(chunk, tg);
    DD_PROF_LOOP_ADD(0);
    if (C::NCHUNK > 1 && tg == C::NTG - 1 && chunk + 1 < C::NCHUNK && !(abl & 1)) {
      if constexpr (C::ONEBUF) {
        // one patch buffer: every wave has taken its last fragment of this chunk out of it before anybody overwrites it
        DD_WAIT_LGKM0();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
      transform_write(chunk + 1, ((chunk + 1) & (C::NPB - 1)) * C::PATCH_BYTES, (par + 1) % RD);
    }
    DD_PROF_LOOP_ADD(1);
    // The next stage's weights (this wave's DMA pieces) must have landed before the barrier.  VMEM ops retire in
    // issue order and the DMA was issued BEFORE this stage's raw patch loads, so when those loads were issued in
    // this stage it is enough to wait until at most NRAW (= the raw loads) are outstanding: they keep flying for
    // two more stages (cdna guide T4: counted vmcnt).  Every wave issues exactly NRAW loads (clamped addresses).
    if (C::NCHUNK > 1 && tg == 0 && chunk + RD < C::NCHUNK && (C::NTG > 1 || RD == 2) && !(abl & (2 | 128))) {
      DD_WAIT_VM_LGKM0(NRAW);
    } else {
      DD_WAIT_VM_LGKM0(0);
    }
    // raw s_barrier: __syncthreads() would make hipcc drain vmcnt(0) because an LDS-DMA may be pending
    if (!(abl & 64)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    DD_PROF_LOOP_ADD(2);
  };

#pragma unroll 1
  for (int sp = 0;; ++sp) {        // the cout splits of this workgroup (one pass unless C::SPW > 1)
  if constexpr (RD == 2) {
    // two raw slots: the chunk loop advances two chunks per trip so that the slot index is a constant
    static_assert(C::NCHUNK % 2 == 0, "even number of channel chunks");
#pragma unroll 1
    for (int chunk = 0; chunk < C::NCHUNK; chunk += 2) {
#pragma unroll
      for (int par = 0; par < 2; ++par) {
        if constexpr (C::TG == 1) {
#pragma unroll
          for (int tg = 0; tg < C::NTG; ++tg) stage(chunk + par, tg, par);
        } else {
#pragma unroll 1
          for (int tg = 0; tg < C::NTG; ++tg) stage(chunk + par, tg, par);
        }
      }
    }
  } else if constexpr (C::TG == 1) {
    // one tap per stage: unroll the nine stages of a chunk so that (dy, dx) are compile-time
#pragma unroll 1
    for (int chunk = 0; chunk < C::NCHUNK; ++chunk) {
#pragma unroll
      for (int tg = 0; tg < C::NTG; ++tg) stage(chunk, tg, 0);
    }
  } else {
#pragma unroll 1
    for (int chunk = 0; chunk < C::NCHUNK; ++chunk) {
#pragma unroll 1
      for (int tg = 0; tg < C::NTG; ++tg) stage(chunk, tg, 0);
    }
  }

  DD_PROF_MARK(3);
  DD_PROF_LOOP_STORE();
  // ---- epilogue: bias, GroupNorm partial sums, store --------------------------------------------------
#ifndef DD_HOST_EMULATION
  if constexpr (C::SPW > 1) asm volatile("" : "+s"(n0));      // keeps the epilogue's store addresses from being hoisted across the MFMA loop (register pressure)
#endif
  constexpr int NG_LOCAL = (C::COUT == COND_C) ? C::NT / (COND_C / GN_GROUPS) : 4;
  float ls[4] = {0.f, 0.f, 0.f, 0.f}, lq[4] = {0.f, 0.f, 0.f, 0.f};
  char* out_b = reinterpret_cast<char*>(p.out) + (size_t)e_b * h * w * C::COUT * C::OUT_ESZ;
  if constexpr (C::IS_NECK) {
    const int cs = p.out_cstride ? p.out_cstride : C::COUT;
    out_b = reinterpret_cast<char*>(p.out) + ((size_t)e_b * cs + p.out_coff) * h * w * C::OUT_ESZ;
  }
#pragma unroll
  for (int m = 0; m < C::WM; ++m) {
    const int gy = e_y0 + wave * C::WM + m, gx = e_x0 + li;
    const bool pvalid = gy < h && gx < w && !(abl & 16);
    int trow = -1;                           // ADD_T: this pixel's row of the E[t] border table (-1: the reference class, already in tab_bias)
    if constexpr (C::ADD_T) {
      const int rc = swin_tt_class(gy, h), cc = swin_tt_class(gx, w);
      if (pvalid && (rc != swin_tt_ref(h) || cc != swin_tt_ref(w))) trow = 1 + rc * SWIN_TT_AX + cc;
    }
    unsigned tapmask = 0x1FFu;               // ADD_C: taps of the 3x3 window that fall inside the image at this pixel
    if constexpr (C::ADD_C) {
      const unsigned ry = (gy >= 1 ? 1u : 0u) | 2u | (gy + 1 < h ? 4u : 0u);
      const unsigned rx = (gx >= 1 ? 1u : 0u) | 2u | (gx + 1 < w ? 4u : 0u);
      tapmask = ((ry & 1u) ? rx : 0u) | ((ry & 2u) ? (rx << 3) : 0u) | ((ry & 4u) ? (rx << 6) : 0u);
    }
    // value (n, q) of this pixel row: accumulator quad + bias (+ the hoisted terms' border corrections, ReLU / FPN addend of the once-per-image layers)
    auto value_of = [&](int n, int q, float (&v)[4]) {
      const int co = n0 + n * 32 + 8 * q + 4 * g;
      (void)co;
        float4 bv = *reinterpret_cast<const float4*>(tab_bias + (sp ^ flip) * C::NT + n * 32 + 8 * q + 4 * g);
        if constexpr (C::ADD_C) {
          // tab_bias already holds bias + the full 9-tap E[t] sum; pixels on the image border take the missing taps out
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
        if constexpr (C::ADD_T) {
          if (trow >= 0) {
            const float4 dv = *reinterpret_cast<const float4*>(p.ttab + (size_t)e_b * p.ttab_bstride + trow * HID_C + n * 32 + 8 * q + 4 * g);
            bv.x += dv.x; bv.y += dv.y; bv.z += dv.z; bv.w += dv.w;
          }
          if constexpr (C::PRED5) {
            // pixels on the image border: take out what the 5x5 form summed through pred.0 taps that leave the image (swin_bcorr)
            if (pvalid && (gy == 0 || gy == h - 1 || gx == 0 || gx == w - 1)) {
              const float* ring = p.bcorr + (size_t)e_b * swin_ring_stride(h, w) * HID_C + n * 32 + 8 * q + 4 * g;
              const float4 cv = *reinterpret_cast<const float4*>(ring + (size_t)swin_ring_index(gy, gx, h, w) * HID_C);
              bv.x -= cv.x; bv.y -= cv.y; bv.z -= cv.z; bv.w -= cv.w;
              if ((gy == 0 || gy == h - 1) && (gx == 0 || gx == w - 1)) {      // corner: + the taps that leave sideways
                const float4 cc = *reinterpret_cast<const float4*>(ring + (size_t)(swin_ring_size(h, w) + (gy ? 2 : 0) + (gx ? 1 : 0)) * HID_C);
                bv.x -= cc.x; bv.y -= cc.y; bv.z -= cc.z; bv.w -= cc.w;
              }
            }
          }
        }
        if constexpr (C::SPLIT) {      // accumulators carry the operands' scales: one exact power-of-two multiply, fused with the bias add
          constexpr float OS = split_oscale(C::PRO);
          v[0] = fmaf(acc[n][m][q * 4 + 0], OS, bv.x); v[1] = fmaf(acc[n][m][q * 4 + 1], OS, bv.y);
          v[2] = fmaf(acc[n][m][q * 4 + 2], OS, bv.z); v[3] = fmaf(acc[n][m][q * 4 + 3], OS, bv.w);
        } else {
          v[0] = acc[n][m][q * 4 + 0] + bv.x; v[1] = acc[n][m][q * 4 + 1] + bv.y;
          v[2] = acc[n][m][q * 4 + 2] + bv.z; v[3] = acc[n][m][q * 4 + 3] + bv.w;
        }
        if constexpr (C::RELU_OUT) {
#pragma unroll
          for (int i = 0; i < 4; ++i) v[i] = (v[i] < 0.f) ? 0.f : v[i];      // (once-per-image layers: torch.relu keeps a NaN, v_max_f32 drops it)
          if constexpr (C::IS_LAT) {
            // FPN top-down term: x = relu(bn(conv(f))) + pooled(up(pre_x))  (reference ...res.py:113-116)
            if (p.addend != nullptr && pvalid) {
              constexpr int AESZ = C::SPLIT ? 4 : C::ESZ;      // the addend is a tensor of this mode's inner layers: the operand kind (split f16: fp32)
              const char* ab = reinterpret_cast<const char*>(p.addend) + ((size_t)e_b * h * w * C::COUT + act_offset(C::COUT, h, w, 0, co, gy, gx)) * AESZ;
              float a4[4];
              if constexpr (AESZ == 4) {
                const float4 t4 = *reinterpret_cast<const float4*>(ab);
                a4[0] = t4.x; a4[1] = t4.y; a4[2] = t4.z; a4[3] = t4.w;
              } else {
                const uint2 t2 = *reinterpret_cast<const uint2*>(ab);
                if constexpr (EK == EK_BF16) {
                  a4[0] = bf16_to_f32(t2.x & 0xFFFFu); a4[1] = bf16_to_f32(t2.x >> 16); a4[2] = bf16_to_f32(t2.y & 0xFFFFu); a4[3] = bf16_to_f32(t2.y >> 16);
                } else {
                  a4[0] = f16_to_f32(t2.x & 0xFFFFu); a4[1] = f16_to_f32(t2.x >> 16); a4[2] = f16_to_f32(t2.y & 0xFFFFu); a4[3] = f16_to_f32(t2.y >> 16);
                }
              }
#pragma unroll
              for (int i = 0; i < 4; ++i) v[i] += a4[i];
            }
          }
        }
    };
    if constexpr (C::Q15) {
      // y3 as int16 with one fp32 scale per pixel (dd_kernels.h, EK_F16R): pass 1 finishes the values in place (statistics from the fp32 values),
      // and finds the pixel's max |.| -- its 64 couts sit in this lane and in lane ^ 32; pass 2 normalises, packs (v_cvt_pknorm_i16_f32) and stores
      float pmax = 0.f;
#pragma unroll
      for (int n = 0; n < C::WN; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
          value_of(n, q, v);
          if (pvalid) {
            const float s = (v[0] + v[1]) + (v[2] + v[3]);
            const float sq = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3])));
            const int lg = 2 * n + (q >> 1);
            ls[lg] += s; lq[lg] += sq;
          }
          pmax = fmaxf(pmax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[n][m][q * 4 + i] = v[i];
        }
      pmax = fmaxf(pmax, __shfl_xor(pmax, 32, 64));
      const float inv = pmax > 0.f ? 1.f / pmax : 0.f;
      if (pvalid && g == 0) p.out_scale[(size_t)e_b * h * w + (size_t)gy * w + gx] = pmax * (1.f / Q15_ONE);
#pragma unroll
      for (int n = 0; n < C::WN; ++n) {
        uint2 pk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          pk[q].x = DD_CVT_PKNORM_I16(acc[n][m][q * 4 + 0] * inv, acc[n][m][q * 4 + 1] * inv);
          pk[q].y = DD_CVT_PKNORM_I16(acc[n][m][q * 4 + 2] * inv, acc[n][m][q * 4 + 3] * inv);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const auto rx = __builtin_amdgcn_permlane32_swap(pk[2 * k].x, pk[2 * k + 1].x, false, false);
          const auto ry = __builtin_amdgcn_permlane32_swap(pk[2 * k].y, pk[2 * k + 1].y, false, false);
          const int co = n0 + n * 32 + 16 * k + 8 * g;
          if (pvalid) *reinterpret_cast<uint4*>(out_b + act_offset(C::COUT, h, w, 0, co, gy, gx) * 2) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
        }
      }
      continue;
    }
#pragma unroll
    for (int n = 0; n < C::WN; ++n) {
      uint2 pk[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (C::COUT < 32 && q >= 2) continue;            // conv4: couts 16..31 are zero padding
        const int co = n0 + n * 32 + 8 * q + 4 * g;
        float v[4];
        value_of(n, q, v);
        if (C::STATS && pvalid) {
          const float s = (v[0] + v[1]) + (v[2] + v[3]);
          const float sq = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3])));
          const int lg = (C::COUT == COND_C) ? (n >> 1) : (C::COUT == HID_C) ? (2 * n + (q >> 1)) : q;
          ls[lg] += s; lq[lg] += sq;
        }
        if constexpr (C::LAYER == 8) {
          // hoisted condition term: fp32, accumulator-fragment order [tile][wave][n][m][q][lane] (read back by layer 9 only)
          const size_t fi = ((((size_t)e_tile * C::WAVES + wave) * C::WN + n) * C::WM + m) * 256 + q * 64 + lane;
          if constexpr (C::CADD16) reinterpret_cast<uint2*>(p.out)[fi] = make_uint2(pack2<EK_F16>(v[0], v[1]), pack2<EK_F16>(v[2], v[3]));
          else reinterpret_cast<float4*>(p.out)[fi] = make_float4(v[0], v[1], v[2], v[3]);
        } else if constexpr (C::OUT_ESZ == 4) {
          if (pvalid && (C::COUT_PAD == C::COUT || co < C::COUT)) {      // (padding couts of a rounded-up tile are never stored)
            if constexpr (C::SCATTER) {
              const int par = co >> 8, cc = co & (COND_C - 1);
              *reinterpret_cast<float4*>(reinterpret_cast<char*>(p.out) + ((size_t)e_b * 4 * h * w * COND_C +
                  act_offset(COND_C, 2 * h, 2 * w, 0, cc, 2 * gy + (par >> 1), 2 * gx + (par & 1))) * 4) = make_float4(v[0], v[1], v[2], v[3]);
            } else {
              *reinterpret_cast<float4*>(out_b + act_offset(C::COUT, h, w, 0, co, gy, gx) * 4) = make_float4(v[0], v[1], v[2], v[3]);
            }
          }
        } else {
          pk[q].x = pack2<C::OUT_K>(v[0], v[1]);
          pk[q].y = pack2<C::OUT_K>(v[2], v[3]);
        }
      }
      if constexpr (C::OUT_ESZ == 2) {
        // half-wave g holds couts 8q+4g..+3: swap so that g=0 lanes own couts 16k..16k+7, g=1 lanes 16k+8..16k+15
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const auto rx = __builtin_amdgcn_permlane32_swap(pk[2 * k].x, pk[2 * k + 1].x, false, false);
          const auto ry = __builtin_amdgcn_permlane32_swap(pk[2 * k].y, pk[2 * k + 1].y, false, false);
          const int co = n0 + n * 32 + 16 * k + 8 * g;
          if (pvalid && (C::COUT_PAD == C::COUT || co < C::COUT)) {
            if constexpr (C::SCATTER) {
              const int par = co >> 8, cc = co & (COND_C - 1);
              *reinterpret_cast<uint4*>(reinterpret_cast<char*>(p.out) + ((size_t)e_b * 4 * h * w * COND_C +
                  act_offset(COND_C, 2 * h, 2 * w, 0, cc, 2 * gy + (par >> 1), 2 * gx + (par & 1))) * 2) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
            } else {
              *reinterpret_cast<uint4*>(out_b + act_offset(C::COUT, h, w, 0, co, gy, gx) * 2) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
            }
          }
        }
      }
    }
  }
  DD_PROF_MARK(4);
  if constexpr (!C::STATS) { DD_PROF_MARK(5); return; }
  // wave-level butterfly in fp32 (64 fp32 lane partials of <= 64 values each; the cross-wave / cross-workgroup
  // accumulation below is fp64).  COUT=16: lanes of half g hold groups {g, 2+g} -> reduce inside each half only.
  constexpr int TOP = (C::COUT < 32) ? 16 : 32;
#pragma unroll
  for (int off = TOP; off >= 1; off >>= 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { ls[k] += __shfl_xor(ls[k], off, 64); lq[k] += __shfl_xor(lq[k], off, 64); }
  }
  double ds[4], dq[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) { ds[k] = (double)ls[k]; dq[k] = (double)lq[k]; }
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
  if (tid < 2 * NG_LOCAL && !(abl & 32)) {
    double tot = 0.0;
#pragma unroll
    for (int wv = 0; wv < C::WAVES; ++wv) tot += s_red[wv * 8 + tid];
    const int gbase = (C::COUT == COND_C) ? (n0 / (COND_C / GN_GROUPS)) : 0;
    double* dst = p.stats_out + ((size_t)e_b * STAT_SLOTS + (wgid % STAT_SLOTS)) * STAT_STRIDE + gbase * 2 + tid;
    atomicAdd(dst, tot);
  }
  DD_PROF_MARK(5);
  if (sp + 1 >= C::SPW) break;
  // next cout split over the same patch: its first weight stage was requested during this split's last stage and has landed
  n0 = (nsplit + ((sp + 1) ^ flip)) * C::NT;
  sbase += C::NSTAGE;
  init_acc();
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
  unsigned n_wg = (unsigned)(p.tiles_x * p.tiles_y * p.B * (C::COUT_PAD / C::NT / C::SPW));
  dim3 grid(n_wg, 1);
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
    case 5: return launch_one2<EK, 5>(p, s);
    case 6: return launch_one2<EK, 6>(p, s);
    case 7: return launch_one2<EK, 7>(p, s);
    case 8: return launch_one2<EK, 8>(p, s);
    case 9: return launch_one2<EK, 9>(p, s);
    case 10: return launch_one2<EK, 10>(p, s);
    case 11: return launch_one2<EK, 11>(p, s);
    case 12: return launch_one2<EK, 12>(p, s);
    case 13: return launch_one2<EK, 13>(p, s);
    case 14: return launch_one2<EK, 14>(p, s);
    case 15: return launch_one2<EK, 15>(p, s);
    case 16: return launch_one2<EK, 16>(p, s);
    case 17: return launch_one2<EK, 17>(p, s);
    case 18: return launch_one2<EK, 18>(p, s);
    case 20: return launch_one2<EK, 20>(p, s);
    case 21: return launch_one2<EK, 21>(p, s);
    case 22: return launch_one2<EK, 22>(p, s);
    case 23: return launch_one2<EK, 23>(p, s);
    case 24: return launch_one2<EK, 24>(p, s);
    case 25: return launch_one2<EK, 25>(p, s);
    case 26: return launch_one2<EK, 26>(p, s);
    case 30: return launch_one2<EK, 30>(p, s);
    case 31: return launch_one2<EK, 31>(p, s);
    case 32: return launch_one2<EK, 32>(p, s);
    case 33: return launch_one2<EK, 33>(p, s);
    case 34: return launch_one2<EK, 34>(p, s);
    case 35: return launch_one2<EK, 35>(p, s);
    case 36: return launch_one2<EK, 36>(p, s);
    case 37: return launch_one2<EK, 37>(p, s);
    case 38: return launch_one2<EK, 38>(p, s);
    case 39: return launch_one2<EK, 39>(p, s);
    case 40: return launch_one2<EK, 40>(p, s);
    case 41: return launch_one2<EK, 41>(p, s);
    case BIG_CONV3C: if constexpr (EK != EK_F32) return launch_one2<EK, BIG_CONV3C>(p, s); else return hipErrorInvalidValue;
    case BIG_CONV3H: if constexpr (EK != EK_F32) return launch_one2<EK, BIG_CONV3H>(p, s); else return hipErrorInvalidValue;
    case ONE_CONV3H: if constexpr (EK == EK_F16) return launch_one2<EK, ONE_CONV3H>(p, s); else return launch_one2<EK, 9>(p, s);      // (instantiated for the kinds the denoiser's modes run)
    case SWIN_CONVA_H: return launch_one2<EK, SWIN_CONVA_H>(p, s);
    case SWIN_PRED_H: return launch_one2<EK, SWIN_PRED_H>(p, s);
    case SWIN_PRED5_H: return launch_one2<EK, SWIN_PRED5_H>(p, s);
    case SWIN_PRED5B_H: if constexpr (EK != EK_F32) return launch_one2<EK, SWIN_PRED5B_H>(p, s); else return hipErrorInvalidValue;
    case 54: return launch_one2<EK, 54>(p, s);
    case 55: return launch_one2<EK, 55>(p, s);
    case 56: return launch_one2<EK, 56>(p, s);
    case 57: return launch_one2<EK, 57>(p, s);
    case 58: return launch_one2<EK, 58>(p, s);
    case 59: return launch_one2<EK, 59>(p, s);
    case 60: return launch_one2<EK, 60>(p, s);
    case 61: return launch_one2<EK, 61>(p, s);
    case 62: return launch_one2<EK, 62>(p, s);
    case 63: return launch_one2<EK, 63>(p, s);
    case 64: return launch_one2<EK, 64>(p, s);
    case 65: return launch_one2<EK, 65>(p, s);
    default: return hipErrorInvalidValue;
  }
}
// EK_BF16M (bf16 operands, f16 storage; dd_kernels.h): the layers that change kind have their own instantiations, the thin / once-per-
// image layers run as f16 kernels, everything else (Swin convB, inner FPN layers, data gradients) as bf16 kernels.
static hipError_t launch_layer2_mixed(int layer, const ConvParams& p, hipStream_t s) {
  switch (layer) {
    case 1: case 4: case 8: return launch_layer2<EK_F16>(layer, p, s);
    case 2: return launch_one2<EK_BF16M, 2>(p, s);
    case 3: return launch_one2<EK_BF16M, 3>(p, s);
    case 5: return launch_one2<EK_BF16M, 5>(p, s);
    case 7: return launch_one2<EK_BF16M, 7>(p, s);
    case 9: return launch_one2<EK_BF16M, 9>(p, s);
    case BIG_CONV3C: return launch_one2<EK_F16, BIG_CONV3C>(p, s);          // (the once-per-image conv3(cond) is an f16 kernel in this mode)
    case BIG_CONV3H: return launch_one2<EK_BF16M, BIG_CONV3H>(p, s);
    case ONE_CONV3H: return launch_one2<EK_BF16M, ONE_CONV3H>(p, s);
    case SWIN_CONVA_H: return launch_one2<EK_BF16M, SWIN_CONVA_H>(p, s);
    case SWIN_PRED_H: return launch_one2<EK_BF16M, SWIN_PRED_H>(p, s);
    case SWIN_PRED5_H: return launch_one2<EK_BF16M, SWIN_PRED5_H>(p, s);
    case SWIN_PRED5B_H: return launch_one2<EK_BF16M, SWIN_PRED5B_H>(p, s);
    case 10: return launch_one2<EK_BF16M, 10>(p, s);
    case 15: return launch_one2<EK_BF16M, 15>(p, s);
    case 24: return launch_one2<EK_BF16M, 24>(p, s);
    default: return launch_layer2<EK_BF16>(layer, p, s);
  }
}
// EK_F16S (split f16): the denoiser's layers and the once-per-image condition layers (FPN, HAHI neck)
static hipError_t launch_layer2_split(int layer, const ConvParams& p, hipStream_t s) {
  switch (layer) {
#define DD_SPLIT_CASE(L) case L: return launch_one2<EK_F16S, L>(p, s);
    DD_SPLIT_CASE(10) DD_SPLIT_CASE(11) DD_SPLIT_CASE(12) DD_SPLIT_CASE(13) DD_SPLIT_CASE(14) DD_SPLIT_CASE(15) DD_SPLIT_CASE(16) DD_SPLIT_CASE(17) DD_SPLIT_CASE(18)
    DD_SPLIT_CASE(24) DD_SPLIT_CASE(25) DD_SPLIT_CASE(26)
    DD_SPLIT_CASE(30) DD_SPLIT_CASE(31) DD_SPLIT_CASE(32) DD_SPLIT_CASE(33) DD_SPLIT_CASE(34) DD_SPLIT_CASE(35) DD_SPLIT_CASE(36) DD_SPLIT_CASE(37) DD_SPLIT_CASE(38) DD_SPLIT_CASE(39) DD_SPLIT_CASE(40) DD_SPLIT_CASE(41)
    DD_SPLIT_CASE(54) DD_SPLIT_CASE(55) DD_SPLIT_CASE(56) DD_SPLIT_CASE(57) DD_SPLIT_CASE(58) DD_SPLIT_CASE(59) DD_SPLIT_CASE(60) DD_SPLIT_CASE(61) DD_SPLIT_CASE(62) DD_SPLIT_CASE(63) DD_SPLIT_CASE(64) DD_SPLIT_CASE(65)
#undef DD_SPLIT_CASE
    case 1: return launch_one2<EK_F16S, 1>(p, s);
    case 2: return launch_one2<EK_F16S, 2>(p, s);
    case 3: return launch_one2<EK_F16S, 3>(p, s);
    case 4: return launch_one2<EK_F16S, 4>(p, s);
    case 5: return launch_one2<EK_F16S, 5>(p, s);
    case 6: return launch_one2<EK_F16S, 6>(p, s);
    case 7: return launch_one2<EK_F16S, 7>(p, s);
    case 8: return launch_one2<EK_F16S, 8>(p, s);
    case 9: case ONE_CONV3H: return launch_one2<EK_F16S, 9>(p, s);
    case CONV3C_NCHW: return launch_one2<EK_F16S, CONV3C_NCHW>(p, s);
    case SWIN_CONVA_H: return launch_one2<EK_F16S, SWIN_CONVA_H>(p, s);
    case SWIN_PRED5_H: return launch_one2<EK_F16S, SWIN_PRED5_H>(p, s);
    default: return hipErrorInvalidValue;
  }
}
// EK_F16R (refined f16): conv1 and the hoisted conv3 have their own instantiations; everything else is the f16 mode's kernel (the split-f16
// layer 8 and the stacked conv4 are launched by dd_api.cpp under their own kinds / launchers)
static hipError_t launch_layer2_refined(int layer, const ConvParams& p, hipStream_t s) {
  switch (layer) {
    case 1: return launch_one2<EK_F16R, 1>(p, s);
    case 9: return launch_one2<EK_F16R, 9>(p, s);
    case BIG_CONV3H: return launch_one2<EK_F16R, BIG_CONV3H>(p, s);
    case ONE_CONV3H: return launch_one2<EK_F16R, ONE_CONV3H>(p, s);
    case SWIN_PRED5_H: return launch_one2<EK_F16R, SWIN_PRED5_H>(p, s);
    case SWIN_PRED5B_H: return launch_one2<EK_F16R, SWIN_PRED5B_H>(p, s);
    default: return launch_layer2<EK_F16>(layer, p, s);
  }
}
hipError_t launch_conv_igemm2(int layer, int ek, const ConvParams& p, hipStream_t s) {
  switch (ek) {
    case EK_F16R: return launch_layer2_refined(layer, p, s);
    case EK_F16S: return launch_layer2_split(layer, p, s);
    case EK_F32: return launch_layer2<EK_F32>(layer, p, s);
    case EK_BF16: return launch_layer2<EK_BF16>(layer, p, s);
    case EK_F16: return launch_layer2<EK_F16>(layer, p, s);
    case EK_BF16M: return launch_layer2_mixed(layer, p, s);
    default: return hipErrorInvalidValue;
  }
}

template <int EK, int LAYER> static PackGeom geom2_of() {
  using C = Cfg2<EK, LAYER>;
  return PackGeom{C::CIN, C::COUT, C::COUT_PAD, C::CK, C::TG, C::NT, C::TH, C::KS, C::NPL, 0};
}
template <int EK> static PackGeom geom2_layer(int layer) {
  switch (layer) {
    case 1: return geom2_of<EK, 1>();
    case 2: return geom2_of<EK, 2>();
    case 3: return geom2_of<EK, 3>();
    case 4: return geom2_of<EK, 4>();
    case 5: return geom2_of<EK, 5>();
    case 6: return geom2_of<EK, 6>();
    case 7: return geom2_of<EK, 7>();
    case 8: return geom2_of<EK, 8>();
    case 9: return geom2_of<EK, 9>();
    case 10: return geom2_of<EK, 10>();
    case 11: return geom2_of<EK, 11>();
    case 12: return geom2_of<EK, 12>();
    case 13: return geom2_of<EK, 13>();
    case 14: return geom2_of<EK, 14>();
    case 15: return geom2_of<EK, 15>();
    case 16: return geom2_of<EK, 16>();
    case 17: return geom2_of<EK, 17>();
    case 18: return geom2_of<EK, 18>();
    case 20: return geom2_of<EK, 20>();
    case 21: return geom2_of<EK, 21>();
    case 22: return geom2_of<EK, 22>();
    case 24: return geom2_of<EK, 24>();
    case 25: return geom2_of<EK, 25>();
    case 26: return geom2_of<EK, 26>();
    case 30: return geom2_of<EK, 30>();
    case 31: return geom2_of<EK, 31>();
    case 32: return geom2_of<EK, 32>();
    case 33: return geom2_of<EK, 33>();
    case 34: return geom2_of<EK, 34>();
    case 35: return geom2_of<EK, 35>();
    case 36: return geom2_of<EK, 36>();
    case 37: return geom2_of<EK, 37>();
    case 38: return geom2_of<EK, 38>();
    case 39: return geom2_of<EK, 39>();
    case 40: return geom2_of<EK, 40>();
    case 41: return geom2_of<EK, 41>();
    case BIG_CONV3C: if constexpr (EK != EK_F32) return geom2_of<EK, BIG_CONV3C>(); else return geom2_of<EK, 8>();      // same packed image as layers 8 / 9, th = 16
    case BIG_CONV3H: if constexpr (EK != EK_F32) return geom2_of<EK, BIG_CONV3H>(); else return geom2_of<EK, 9>();
    case ONE_CONV3H: return geom2_of<EK, 9>();
    case SWIN_CONVA_H: return geom2_of<EK, 5>();          // same packed images as layers 5 / 7
    case SWIN_PRED_H: return geom2_of<EK, 7>();
    case SWIN_PRED5_H: return geom2_of<EK, SWIN_PRED5_H>();
    case SWIN_PRED5B_H: if constexpr (EK != EK_F32) return geom2_of<EK, SWIN_PRED5B_H>(); else return geom2_of<EK, SWIN_PRED5_H>();      // same packed image, th = 16
    case 54: return geom2_of<EK, 54>();
    case 55: return geom2_of<EK, 55>();
    case 56: return geom2_of<EK, 56>();
    case 57: return geom2_of<EK, 57>();
    case 58: return geom2_of<EK, 58>();
    case 59: return geom2_of<EK, 59>();
    case 60: return geom2_of<EK, 60>();
    case 61: return geom2_of<EK, 61>();
    case 62: return geom2_of<EK, 62>();
    case 63: return geom2_of<EK, 63>();
    case 64: return geom2_of<EK, 64>();
    case 65: return geom2_of<EK, 65>();
    default: return geom2_of<EK, 23>();
  }
}
static PackGeom geom2_layer_split(int layer) {
  switch (layer) {
#define DD_SPLIT_CASE(L) case L: return geom2_of<EK_F16S, L>();
    DD_SPLIT_CASE(10) DD_SPLIT_CASE(11) DD_SPLIT_CASE(12) DD_SPLIT_CASE(13) DD_SPLIT_CASE(14) DD_SPLIT_CASE(15) DD_SPLIT_CASE(16) DD_SPLIT_CASE(17) DD_SPLIT_CASE(18)
    DD_SPLIT_CASE(24) DD_SPLIT_CASE(25) DD_SPLIT_CASE(26)
    DD_SPLIT_CASE(30) DD_SPLIT_CASE(31) DD_SPLIT_CASE(32) DD_SPLIT_CASE(33) DD_SPLIT_CASE(34) DD_SPLIT_CASE(35) DD_SPLIT_CASE(36) DD_SPLIT_CASE(37) DD_SPLIT_CASE(38) DD_SPLIT_CASE(39) DD_SPLIT_CASE(40) DD_SPLIT_CASE(41)
    DD_SPLIT_CASE(54) DD_SPLIT_CASE(55) DD_SPLIT_CASE(56) DD_SPLIT_CASE(57) DD_SPLIT_CASE(58) DD_SPLIT_CASE(59) DD_SPLIT_CASE(60) DD_SPLIT_CASE(61) DD_SPLIT_CASE(62) DD_SPLIT_CASE(63) DD_SPLIT_CASE(64) DD_SPLIT_CASE(65)
#undef DD_SPLIT_CASE
    case 1: return geom2_of<EK_F16S, 1>();
    case 2: return geom2_of<EK_F16S, 2>();
    case 3: return geom2_of<EK_F16S, 3>();
    case 4: return geom2_of<EK_F16S, 4>();
    case 5: return geom2_of<EK_F16S, 5>();
    case 6: return geom2_of<EK_F16S, 6>();
    case 7: return geom2_of<EK_F16S, 7>();
    case 8: case CONV3C_NCHW: return geom2_of<EK_F16S, 8>();
    case SWIN_CONVA_H: return geom2_of<EK_F16S, 5>();
    case SWIN_PRED5_H: return geom2_of<EK_F16S, SWIN_PRED5_H>();
    default: return geom2_of<EK_F16S, 9>();
  }
}
PackGeom conv_pack_geom2(int layer, int ek) {
  switch (ek) {
    case EK_F16R: {      // conv1: the split image; conv4: the f16 geometry with the lo halves stacked into the padding cout rows; else f16
      if (layer == 1) return geom2_of<EK_F16S, 1>();
      PackGeom g = geom2_layer<EK_F16>(layer);
      if (layer == 4) g.stack = 1;
      return g;
    }
    case EK_F16S: return geom2_layer_split(layer);
    case EK_F32: return geom2_layer<EK_F32>(layer);
    case EK_BF16: case EK_BF16M: return geom2_layer<EK_BF16>(layer);     // 2-byte kinds share one geometry
    default: return geom2_layer<EK_F16>(layer);
  }
}

}  // namespace dd
