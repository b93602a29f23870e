// dd_dcn.hip -- NLSPN spatial propagation and the DCNv2 operator under it (include/ddepth_dcn.h).
//
// Reference: the vendored DCNv2 CUDA extension (src/model/deformconv/src/cuda/modulated_deform_im2col_cuda.cuh,
// modulated_deform_conv_cuda.cu) and its one caller, NLSPN (src/model/nlspnmodel.py:22-207).  The reference materialises a
// column buffer (C*kh*kw x B*Ho*Wo floats) with one kernel and contracts it with cuBLAS; NLSPN then calls that pair 18 + 8 times
// per image on a ONE-channel map with an all-ones 3x3 "weight".  None of that is a GEMM worth a matrix core: it is a gather.
// Here sampling and contraction are one kernel, nothing but the operands and the result touches HBM, and the two NLSPN stages
// have their own fused kernels:
//   nlspn_affinity_kernel   the whole of _get_offset_affinity after its convolution (24 planes in, 27 out, 8 confidence gathers)
//   nlspn_prop_kernel       one propagation iteration: 18 offset + 9 affinity planes streamed with 16-B loads (4 pixels per lane),
//                           36 bilinear corner gathers per pixel served by L2 / the vector L1 (the depth map is 1.7 MB at KITTI
//                           size), one 16-B store.  112 B of HBM traffic per pixel per iteration: the HBM roofline bounds it.
// All kernels are fp32 (the reference's arithmetic) and follow the reference's evaluation order inside a sample.
#include "../../include/ddepth.h"
#include "../../include/ddepth_dcn.h"

#include <hip/hip_runtime.h>

#include "dd_gcn.h"

#include <climits>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string>

namespace {

thread_local std::string g_dcn_err;

int dcn_fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_dcn_err = buf;
  return code;
}

#define DCN_HIP(expr)                                                                            \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) return dcn_fail(DD_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e));   \
  } while (0)

// ---- bilinear cell of one sampling position (mdmcn_im2col_bilinear, ...cuh:23-54, and the kernel's range test, ...cuh:179) ----
struct Tap {
  float h, w;      // sampling position
  float lh, lw;    // fractional parts
  int hl, wl;      // floor cell
  bool inside;     // h > -1 && w > -1 && h < H && w < W: outside this the sample (and every gradient through it) is zero
};

__device__ __forceinline__ Tap make_tap(float h, float w, int H, int W) {
  Tap t;
  t.h = h;
  t.w = w;
  t.inside = (h > -1.f) && (w > -1.f) && (h < (float)H) && (w < (float)W);
  const float fh = floorf(h), fw = floorf(w);
  t.hl = (int)fh;
  t.wl = (int)fw;
  t.lh = h - fh;
  t.lw = w - fw;
  return t;
}

// the four corner values, zero where a corner lies outside the image (...cuh:36-47)
__device__ __forceinline__ void corner_vals(const float* __restrict__ im, int H, int W, const Tap& t, float& v1, float& v2,
                                            float& v3, float& v4) {
  const bool y0 = t.inside && t.hl >= 0, y1 = t.inside && t.hl + 1 <= H - 1;
  const bool x0 = t.wl >= 0, x1 = t.wl + 1 <= W - 1;
  const float* r = im + (long long)t.hl * W + t.wl;
  v1 = (y0 && x0) ? r[0] : 0.f;
  v2 = (y0 && x1) ? r[1] : 0.f;
  v3 = (y1 && x0) ? r[W] : 0.f;
  v4 = (y1 && x1) ? r[W + 1] : 0.f;
}

__device__ __forceinline__ float bilerp(const Tap& t, float v1, float v2, float v3, float v4) {
  const float hh = 1.f - t.lh, hw = 1.f - t.lw;                        // ...cuh:33-34
  const float w1 = hh * hw, w2 = hh * t.lw, w3 = t.lh * hw, w4 = t.lh * t.lw;   // ...cuh:50
  return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;                        // ...cuh:52
}

__device__ __forceinline__ float sample_plane(const float* __restrict__ im, int H, int W, float h, float w) {
  const Tap t = make_tap(h, w, H, W);
  float v1, v2, v3, v4;
  corner_vals(im, H, W, t, v1, v2, v3, v4);
  return bilerp(t, v1, v2, v3, v4);
}

// The same sample with two UNCONDITIONAL 8-byte loads (global_load_dwordx2 at a 4-byte-aligned address) instead of four predicated
// dword loads: row / column indices are clamped into the image, out-of-image corners are zeroed by selects.  Needs W >= 2.
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));
__device__ __forceinline__ float sample_pair(const float* __restrict__ im, int H, int W, float h, float w) {
  const Tap t = make_tap(h, w, H, W);
  const int c0 = min(max(t.wl, 0), W - 2);
  const int r0 = min(max(t.hl, 0), H - 1), r1 = min(max(t.hl + 1, 0), H - 1);
  const f2u p = *reinterpret_cast<const f2u*>(im + (size_t)r0 * W + c0);
  const f2u q = *reinterpret_cast<const f2u*>(im + (size_t)r1 * W + c0);
  const int d = t.wl - c0;                  // inside the range test: -1 (left column out), 0, +1 (right column out)
  const bool y0 = t.inside && t.hl >= 0, y1 = t.inside && t.hl + 1 <= H - 1;
  const float pl = d == 0 ? p.x : (d == 1 ? p.y : 0.f), pr = d == 0 ? p.y : (d == -1 ? p.x : 0.f);
  const float ql = d == 0 ? q.x : (d == 1 ? q.y : 0.f), qr = d == 0 ? q.y : (d == -1 ? q.x : 0.f);
  return bilerp(t, y0 ? pl : 0.f, y0 ? pr : 0.f, y1 ? ql : 0.f, y1 ? qr : 0.f);
}

template <int VEC> struct VecLoad;
template <> struct VecLoad<1> {
  static __device__ __forceinline__ void ld(const float* p, float (&v)[1]) { v[0] = *p; }
  static __device__ __forceinline__ void st(float* p, const float (&v)[1]) { *p = v[0]; }
};
template <> struct VecLoad<4> {
  static __device__ __forceinline__ void ld(const float* p, float (&v)[4]) {
    const float4 q = *reinterpret_cast<const float4*>(p);
    v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
  }
  static __device__ __forceinline__ void st(float* p, const float (&v)[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  }
};

// ------------------------------------------------------------------------------------------------------------------------
// One propagation iteration of NLSPN (nlspnmodel.py:165-171 via ...cuh:126-194 with C = 1 and the 1 x K "GEMM" of
// modulated_deform_conv_cuda.cu:107-115):  out = b + sum_k w_k * aff_k * bilinear(fin, pixel + tap_k + offset_k).
// Block = 64 x 4 lanes, a lane owns VEC consecutive pixels of a row: the 27 operand planes are read as 16-B row segments, the
// result is one 16-B store; `blend` (preserve_input) additionally receives the NEXT iteration's input
// (1 - mask_fix) * out + mask_fix * feat_fix (nlspnmodel.py:199-201), so that the next launch gathers from one array.
// ------------------------------------------------------------------------------------------------------------------------
template <int KF, int VEC, bool PAIR>
__global__ void __launch_bounds__(256) nlspn_prop_kernel(const float* __restrict__ fin, const float* __restrict__ offset,
                                                         const float* __restrict__ aff, const float* __restrict__ wk,
                                                         const float* __restrict__ bk, float* __restrict__ out,
                                                         const float* __restrict__ fix, float* __restrict__ blend, int H, int W) {
  constexpr int K = KF * KF, PAD = (KF - 1) / 2;
  const int b = blockIdx.z;
  const int h = blockIdx.y * 4 + threadIdx.y;
  const int w0 = (blockIdx.x * 64 + threadIdx.x) * VEC;
  if (h >= H || w0 >= W) return;
  const size_t HW = (size_t)H * W;
  const size_t pix = (size_t)h * W + w0;
  const float* im = fin + (size_t)b * HW;
  const float* ob = offset + (size_t)b * 2 * K * HW + pix;
  const float* ab = aff + (size_t)b * K * HW + pix;
  float acc[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) acc[v] = 0.f;
  // one kernel row (KF taps x VEC pixels = 4*KF*VEC gathers) in flight per lane at a time: ~100 VGPRs, 4-5 waves per SIMD, so that
  // the waves' load -> gather -> store phases overlap (fully unrolled the compiler takes 256 VGPRs = 1 wave per SIMD)
#pragma unroll 1
  for (int i = 0; i < KF; ++i) {
#pragma unroll
    for (int j = 0; j < KF; ++j) {
      const int k = i * KF + j;
      float oh[VEC], ow[VEC], a[VEC];
      VecLoad<VEC>::ld(ob + (size_t)(2 * k) * HW, oh);
      VecLoad<VEC>::ld(ob + (size_t)(2 * k + 1) * HW, ow);
      VecLoad<VEC>::ld(ab + (size_t)k * HW, a);
      const float wgt = wk[k];
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const float hs = (float)(h - PAD + i) + oh[v];             // ...cuh:177: h_in + i * dilation_h + offset_h
        const float ws = (float)(w0 + v - PAD + j) + ow[v];
        const float val = PAIR ? sample_pair(im, H, W, hs, ws) : sample_plane(im, H, W, hs, ws);
        acc[v] += (val * a[v]) * wgt;                              // col = val * mask (...cuh:189), then . w
      }
    }
  }
  const float bias = bk[0];
  float res[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) res[v] = bias + acc[v];
  VecLoad<VEC>::st(out + (size_t)b * HW + pix, res);
  if (blend != nullptr) {
    float fx[VEC];
    VecLoad<VEC>::ld(fix + (size_t)b * HW + pix, fx);
#pragma unroll
    for (int v = 0; v < VEC; ++v) fx[v] = fx[v] > 0.f ? fx[v] : res[v];
    VecLoad<VEC>::st(blend + (size_t)b * HW + pix, fx);
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// The same iteration with the gathered map staged through LDS.  A block owns a TH x 64 pixel tile and first copies the tile plus an
// R-pixel halo of `fin` into LDS, ZERO-filled outside the image -- which is exactly the reference's corner rule (...cuh:36-47), so
// samples whose 2x2 cell lies in the staged window need no validity logic at all: two ds_read2_b32 and the four weights.  Samples
// that leave the window (|offset| beyond ~R pixels) fall back to the global-memory sampler.  A lane owns ONE pixel per row
// (consecutive lanes = consecutive columns: at small offsets the LDS reads are conflict-free; row stride 81 dwords) and walks TH / 4 rows.
// The 27 operand planes are streamed with coalesced dword loads (256 B per wave instruction).
// ------------------------------------------------------------------------------------------------------------------------
template <int KF, int TH, bool PAIR>
__global__ void __launch_bounds__(256) nlspn_prop_lds_kernel(const float* __restrict__ fin, const float* __restrict__ offset,
                                                             const float* __restrict__ aff, const float* __restrict__ wk,
                                                             const float* __restrict__ bk, float* __restrict__ out,
                                                             const float* __restrict__ fix, float* __restrict__ blend, int H, int W) {
  constexpr int K = KF * KF, PAD = (KF - 1) / 2, R = 8, TW = 64, LW = TW + 2 * R, LH = TH + 2 * R, LS = LW + 1;
  __shared__ float tile[LH * LS];
  const int b = blockIdx.z;
  const int y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
  const size_t HW = (size_t)H * W;
  const float* im = fin + (size_t)b * HW;
  const int tid = threadIdx.x;
  {
    constexpr int NFILL = (LH * LW + 255) / 256;      // all window loads in flight before the first LDS store
    float tv[NFILL];
#pragma unroll
    for (int it = 0; it < NFILL; ++it) {
      const int idx = tid + it * 256;
      const int r = idx / LW, c = idx - r * LW;
      const int gy = y0 - R + r, gx = x0 - R + c;
      tv[it] = im[(size_t)min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1)];      // clamped: always a valid address
    }
    // pin the loads above the selects below (otherwise LLVM sinks each load into its "in the image" branch, one latency after another)
#pragma unroll
    for (int it = 0; it < NFILL; ++it) DD_PIN_VGPR(tv[it]);
#pragma unroll
    for (int it = 0; it < NFILL; ++it) {
      const int idx = tid + it * 256;
      const int r = idx / LW, c = idx - r * LW;
      const int gy = y0 - R + r, gx = x0 - R + c;
      if (idx < LH * LW) tile[r * LS + c] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? tv[it] : 0.f;
    }
  }
  __syncthreads();
  const int tx = tid & 63, ty = tid >> 6;
  const int w = x0 + tx;
  if (w >= W) return;
  const float bias = bk[0];
  for (int rr = ty; rr < TH; rr += 4) {
    const int h = y0 + rr;
    if (h >= H) break;
    const size_t pix = (size_t)h * W + w;
    const float* ob = offset + (size_t)b * 2 * K * HW + pix;
    const float* ab = aff + (size_t)b * K * HW + pix;
    float acc = 0.f;
    // RA kernel rows at a time: first ALL their operand loads (3 * RA * KF independent dword loads in flight), then the LDS gathers
    // (unconditional, indices clamped into the window), then -- rarely -- the global fallback for samples outside the window
    constexpr int RA = KF == 3 ? 3 : 1;
#pragma unroll 1
    for (int i0 = 0; i0 < KF; i0 += RA) {
      float oh[RA * KF], ow[RA * KF], a[RA * KF];
#pragma unroll
      for (int q = 0; q < RA * KF; ++q) {
        const int k = i0 * KF + q;
        oh[q] = ob[(size_t)(2 * k) * HW];
        ow[q] = ob[(size_t)(2 * k + 1) * HW];
        a[q] = ab[(size_t)k * HW];
      }
#pragma unroll
      for (int q = 0; q < RA * KF; ++q) {
        const int i = i0 + q / KF, j = q % KF;
        const float hs = (float)(h - PAD + i) + oh[q];
        const float ws = (float)(w - PAD + j) + ow[q];
        const float fh = floorf(hs), fw = floorf(ws);
        const int ly = (int)fh - (y0 - R), lx = (int)fw - (x0 - R);
        const bool in_win = (unsigned)ly < (unsigned)(LH - 1) && (unsigned)lx < (unsigned)(LW - 1);
        const float* p = tile + min(max(ly, 0), LH - 2) * LS + min(max(lx, 0), LW - 2);
        const float lh = hs - fh, lw = ws - fw, hh = 1.f - lh, hw = 1.f - lw;
        float val = (hh * hw) * p[0] + (hh * lw) * p[1] + (lh * hw) * p[LS] + (lh * lw) * p[LS + 1];
        if (!in_win) val = PAIR ? sample_pair(im, H, W, hs, ws) : sample_plane(im, H, W, hs, ws);
        acc += (val * a[q]) * wk[i * KF + j];
      }
    }
    const float res = bias + acc;
    out[(size_t)b * HW + pix] = res;
    if (blend != nullptr) {
      const float fx = fix[(size_t)b * HW + pix];
      blend[(size_t)b * HW + pix] = fx > 0.f ? fx : res;
    }
  }
}

// 3x3 specialisation of the LDS kernel with the operand stream software-pipelined across the TH / 4 rows a lane walks: the 27 operand
// loads of the first row are issued BEFORE the window fill and the barrier (HBM latency overlaps the L2 fill), and each row's loads
// are issued before the previous row is evaluated (two register sets).
template <int TH, bool PAIR>
__global__ void __launch_bounds__(256) nlspn_prop_lds3_kernel(const float* __restrict__ fin, const float* __restrict__ offset,
                                                              const float* __restrict__ aff, const float* __restrict__ wk,
                                                              const float* __restrict__ bk, float* __restrict__ out,
                                                              const float* __restrict__ fix, float* __restrict__ blend, int H, int W) {
  constexpr int KF = 3, K = 9, PAD = 1, R = 8, TW = 64, LW = TW + 2 * R, LH = TH + 2 * R, LS = LW + 1, NR = TH / 4;
  __shared__ float tile[LH * LS];
  const int b = blockIdx.z;
  const int y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
  const size_t HW = (size_t)H * W;
  const float* im = fin + (size_t)b * HW;
  const int tid = threadIdx.x;
  const int tx = tid & 63, ty = tid >> 6;
  const int w = x0 + tx;
  const int wc = min(w, W - 1);                       // clamped: lanes past the right edge load (and discard) valid addresses
  const float* ob = offset + (size_t)b * 2 * K * HW;
  const float* ab = aff + (size_t)b * K * HW;
  float ops[2][3 * K];                                // [set][oh(9), ow(9), a(9)]
  auto load_ops = [&](int n, float (&o)[3 * K]) {
    const size_t pix = (size_t)min(y0 + ty + 4 * n, H - 1) * W + wc;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      o[k] = ob[(size_t)(2 * k) * HW + pix];
      o[K + k] = ob[(size_t)(2 * k + 1) * HW + pix];
      o[2 * K + k] = ab[(size_t)k * HW + pix];
    }
  };
  load_ops(0, ops[0]);
  {
    constexpr int NFILL = (LH * LW + 255) / 256;
    float tv[NFILL];
#pragma unroll
    for (int it = 0; it < NFILL; ++it) {
      const int idx = tid + it * 256;
      const int r = idx / LW, c = idx - r * LW;
      tv[it] = im[(size_t)min(max(y0 - R + r, 0), H - 1) * W + min(max(x0 - R + c, 0), W - 1)];
    }
#pragma unroll
    for (int it = 0; it < NFILL; ++it) DD_PIN_VGPR(tv[it]);      // keep the loads above the selects (see the generic kernel)
#pragma unroll
    for (int it = 0; it < NFILL; ++it) {
      const int idx = tid + it * 256;
      const int r = idx / LW, c = idx - r * LW;
      const int gy = y0 - R + r, gx = x0 - R + c;
      if (idx < LH * LW) tile[r * LS + c] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? tv[it] : 0.f;
    }
  }
  __syncthreads();
  const float bias = bk[0];
  float wt[K];
#pragma unroll
  for (int k = 0; k < K; ++k) wt[k] = wk[k];
#pragma unroll
  for (int n = 0; n < NR; ++n) {
    if (n + 1 < NR) load_ops(n + 1, ops[(n + 1) & 1]);
    const float(&o)[3 * K] = ops[n & 1];
    const int h = y0 + ty + 4 * n;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int i = k / KF, j = k % KF;
      const float hs = (float)(h - PAD + i) + o[k];
      const float ws = (float)(w - PAD + j) + o[K + k];
      const float fh = floorf(hs), fw = floorf(ws);
      const int ly = (int)fh - (y0 - R), lx = (int)fw - (x0 - R);
      const bool in_win = (unsigned)ly < (unsigned)(LH - 1) && (unsigned)lx < (unsigned)(LW - 1);
      const float* p = tile + min(max(ly, 0), LH - 2) * LS + min(max(lx, 0), LW - 2);
      const float lh = hs - fh, lw = ws - fw, hh = 1.f - lh, hw = 1.f - lw;
      float val = (hh * hw) * p[0] + (hh * lw) * p[1] + (lh * hw) * p[LS] + (lh * lw) * p[LS + 1];
      if (!in_win) val = PAIR ? sample_pair(im, H, W, hs, ws) : sample_plane(im, H, W, hs, ws);
      acc += (val * o[2 * K + k]) * wt[k];
    }
    if (h < H && w < W) {
      const size_t pix = (size_t)b * HW + (size_t)h * W + w;
      const float res = bias + acc;
      out[pix] = res;
      if (blend != nullptr) {
        const float fx = fix[pix];
        blend[pix] = fx > 0.f ? fx : res;
      }
    }
  }
}

// dst = (1 - mask_fix) * src + mask_fix * fix with mask_fix = fix > 0 (the blend before the FIRST iteration)
__global__ void __launch_bounds__(256) nlspn_blend_kernel(const float* __restrict__ src, const float* __restrict__ fix,
                                                          float* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) {
    const float f = fix[i];
    dst[i] = f > 0.f ? f : src[i];
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// NLSPN._get_offset_affinity after conv_offset_aff (nlspnmodel.py:90-163), one lane per pixel.
// ------------------------------------------------------------------------------------------------------------------------
template <int KF>
__global__ void __launch_bounds__(256) nlspn_affinity_kernel(const float* __restrict__ raw, const float* __restrict__ conf,
                                                             const float* __restrict__ gamma_p, const float* __restrict__ wconf_p,
                                                             const float* __restrict__ bconf_p, float* __restrict__ offset,
                                                             float* __restrict__ aff, int H, int W, int mode, int legacy) {
  constexpr int K = KF * KF, NUM = K - 1, REF = NUM / 2;
  constexpr float CEN = (float)((KF - 1) / 2);
  const int b = blockIdx.z;
  const int h = blockIdx.y * 4 + threadIdx.y;
  const int w = blockIdx.x * 64 + threadIdx.x;
  if (h >= H || w >= W) return;
  const size_t HW = (size_t)H * W;
  const size_t pix = (size_t)h * W + w;
  const float* r = raw + (size_t)b * 3 * NUM * HW + pix;
  float* o = offset + (size_t)b * 2 * K * HW + pix;
  float* ao = aff + (size_t)b * K * HW + pix;
  const float gamma = gamma_p[0];
  const float* cim = conf ? conf + (size_t)b * HW : nullptr;
  const float wconf = conf ? wconf_p[0] : 0.f, bconf = conf ? bconf_p[0] : 0.f;
  float a[NUM];
  float sabs = 0.f;
#pragma unroll
  for (int n = 0; n < NUM; ++n) {
    const int m = n < REF ? n : n + 1;           // position of neighbour n once the zero reference offset is inserted (:96-99)
    // torch.cat((o1, o2), 1).view(B, num, 2, H, W): neighbour n owns channels 2n (rows) and 2n+1 (columns) of the first 2*num (:92-95)
    float oh = r[(size_t)(2 * n) * HW], ow = r[(size_t)(2 * n + 1) * HW];
    if (conf && legacy) {                        // :126-134, in place on a view of `offset`: the returned offsets carry the shift
      oh = (oh + (float)(m / KF)) - CEN;
      ow = (ow + (float)(m % KF)) - CEN;
    }
    o[(size_t)(2 * m) * HW] = oh;
    o[(size_t)(2 * m + 1) * HW] = ow;
    float v = r[(size_t)(2 * NUM + n) * HW];
    if (mode == DD_AFF_TC) v = tanhf(v) / gamma;                        // :103-104
    else if (mode == DD_AFF_TGASS) v = tanhf(v) / (gamma + 1e-8f);      // :105-106
    if (conf) {
      // 1x1 modulated deformable conv of the confidence with a mask of ones, padding 0 (:136-141): bilinear sample at pixel + offset
      const float c = bconf + (sample_plane(cim, H, W, (float)h + oh, (float)w + ow) * 1.0f) * wconf;
      v *= c;                                                           // :143-144
    }
    a[n] = v;
    sabs += fabsf(v);
  }
  o[(size_t)(2 * REF) * HW] = 0.f;
  o[(size_t)(2 * REF + 1) * HW] = 0.f;
  float s = sabs + 1e-4f;                                               // :147-148
  if (mode == DD_AFF_ASS || mode == DD_AFF_TGASS) s = s < 1.f ? 1.f : s;   // :150-151
  float sum = 0.f;
#pragma unroll
  for (int n = 0; n < NUM; ++n) {
    if (mode != DD_AFF_TC) a[n] = a[n] / s;                             // :153-154
    sum += a[n];
    ao[(size_t)(n < REF ? n : n + 1) * HW] = a[n];
  }
  ao[(size_t)REF * HW] = 1.f - sum;                                     // :156-161
}

// ------------------------------------------------------------------------------------------------------------------------
// The same stage INCLUDING its convolution  offset_aff = conv_offset_aff(guidance)  (nlspnmodel.py:90: Conv2d(ch_g, 3*num, 3, padding 1,
// bias)) for the shipped geometry ch_g = 8, k_g = 3, k_f = 3 (NLSPNModel, nlspnmodel.py:215,287-288): the 24-plane intermediate
// (192 B per pixel written and re-read) never exists.  A block owns an 8 x 64 pixel tile; the 8-channel guidance tile + 1-pixel zero
// halo (the conv's zero padding) is staged in LDS; a lane holds its pixel's 72-value patch in registers and walks the 24 output
// channels with the weights as SCALAR operands (OIHW: the 72 weights of one output channel are contiguous -> s_load_dwordx8).
// ------------------------------------------------------------------------------------------------------------------------
template <int CG>
__global__ void __launch_bounds__(256) nlspn_guide_affinity_kernel(const float* __restrict__ guide, const float* __restrict__ cw,
                                                                   const float* __restrict__ cb, const float* __restrict__ conf,
                                                                   const float* __restrict__ gamma_p, const float* __restrict__ wconf_p,
                                                                   const float* __restrict__ bconf_p, float* __restrict__ offset,
                                                                   float* __restrict__ aff, int H, int W, int mode, int legacy) {
  constexpr int KF = 3, K = 9, NUM = 8, REF = 4, TH = 8, TW = 64, LW = TW + 2, LH = TH + 2, LS = LW + 1;
  __shared__ float tile[CG * LH * LS];
  __shared__ float asave[NUM * 256];
  const int b = blockIdx.z;
  const int y0 = blockIdx.y * TH, x0 = blockIdx.x * TW;
  const size_t HW = (size_t)H * W;
  const int tid = threadIdx.x;
  {
    constexpr int NFILL = (CG * LH * LW + 255) / 256;
    float tv[NFILL];
#pragma unroll
    for (int it = 0; it < NFILL; ++it) {
      const int idx = tid + it * 256;
      const int c = min(idx / (LH * LW), CG - 1), rem = idx % (LH * LW);
      const int r = rem / LW, col = rem - r * LW;
      tv[it] = guide[((size_t)b * CG + c) * HW + (size_t)min(max(y0 - 1 + r, 0), H - 1) * W + min(max(x0 - 1 + col, 0), W - 1)];
    }
#pragma unroll
    for (int it = 0; it < NFILL; ++it) DD_PIN_VGPR(tv[it]);      // loads stay above the selects (see nlspn_prop_lds_kernel)
#pragma unroll
    for (int it = 0; it < NFILL; ++it) {
      const int idx = tid + it * 256;
      const int c = idx / (LH * LW), rem = idx % (LH * LW);
      const int r = rem / LW, col = rem - r * LW;
      const int gy = y0 - 1 + r, gx = x0 - 1 + col;
      if (idx < CG * LH * LW) tile[(c * LH + r) * LS + col] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? tv[it] : 0.f;
    }
  }
  __syncthreads();
  const int tx = tid & 63, ty = tid >> 6;
  const int w = x0 + tx;
  if (w >= W) return;
  const float gamma = gamma_p[0];
  const float* cim = conf ? conf + (size_t)b * HW : nullptr;
  const float wconf = conf ? wconf_p[0] : 0.f, bconf = conf ? bconf_p[0] : 0.f;
  for (int rr = ty; rr < TH; rr += 4) {
    const int h = y0 + rr;
    if (h >= H) break;
    float g[CG * K];
#pragma unroll
    for (int c = 0; c < CG; ++c)
#pragma unroll
      for (int t = 0; t < K; ++t) g[c * K + t] = tile[(c * LH + rr + t / KF) * LS + tx + t % KF];
    auto conv_out = [&](int co) {                       // one output channel of conv_offset_aff at this pixel; weights are scalar operands
      asm volatile("" ::: "memory");                    // keep this channel's 72 scalar weight loads behind the previous channel's (SGPR budget)
      float p0 = cb[co], p1 = 0.f, p2 = 0.f, p3 = 0.f;  // four independent chains: a single 72-long FMA chain is latency-bound
#pragma unroll
      for (int q = 0; q < CG * K; q += 4) {
        p0 += cw[co * CG * K + q] * g[q];
        p1 += cw[co * CG * K + q + 1] * g[q + 1];
        p2 += cw[co * CG * K + q + 2] * g[q + 2];
        p3 += cw[co * CG * K + q + 3] * g[q + 3];
      }
      return (p0 + p1) + (p2 + p3);
    };
    // ---- from here on: nlspn_affinity_kernel<3> (nlspnmodel.py:92-161), neighbour by neighbour so that only a[] stays live ----
    const size_t pix = (size_t)h * W + w;
    float* o = offset + (size_t)b * 2 * K * HW + pix;
    float* ao = aff + (size_t)b * K * HW + pix;
    // the neighbour loop is NOT unrolled: unrolled, the 1728 scalar weight loads are hoisted together and spill (1000+ SGPRs); the
    // eight affinities wait for the normalisation in a per-lane LDS slot instead of a dynamically indexed register array
    float sabs = 0.f;
#pragma unroll 1
    for (int n = 0; n < NUM; ++n) {
      const int m = n < REF ? n : n + 1;
      float oh = conv_out(2 * n), ow = conv_out(2 * n + 1);
      if (conf && legacy) {
        oh = (oh + (float)(m / KF)) - 1.f;
        ow = (ow + (float)(m % KF)) - 1.f;
      }
      o[(size_t)(2 * m) * HW] = oh;
      o[(size_t)(2 * m + 1) * HW] = ow;
      float v = conv_out(2 * NUM + n);
      if (mode == DD_AFF_TC) v = tanhf(v) / gamma;
      else if (mode == DD_AFF_TGASS) v = tanhf(v) / (gamma + 1e-8f);
      if (conf) v *= bconf + (sample_plane(cim, H, W, (float)h + oh, (float)w + ow) * 1.0f) * wconf;
      asave[n * 256 + tid] = v;
      sabs += fabsf(v);
    }
    o[(size_t)(2 * REF) * HW] = 0.f;
    o[(size_t)(2 * REF + 1) * HW] = 0.f;
    float s_ = sabs + 1e-4f;
    if (mode == DD_AFF_ASS || mode == DD_AFF_TGASS) s_ = s_ < 1.f ? 1.f : s_;
    float sum = 0.f;
#pragma unroll
    for (int n = 0; n < NUM; ++n) {
      float an = asave[n * 256 + tid];                  // written by this lane only: no barrier needed
      if (mode != DD_AFF_TC) an = an / s_;
      sum += an;
      ao[(size_t)(n < REF ? n : n + 1) * HW] = an;
    }
    ao[(size_t)REF * HW] = 1.f - sum;
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// General DCNv2 (any C, groups, deformable groups, stride, dilation).  No model in the reference tree uses it beyond C = 1, so
// these are plain one-lane-per-output kernels: correct for every shape the extension accepts, tuned for none.
// ------------------------------------------------------------------------------------------------------------------------
struct DcnShape {
  int B, C, H, W, Co, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, group, dg;
};

__global__ void __launch_bounds__(256) dcn_fwd_kernel(DcnShape s, const float* __restrict__ in, const float* __restrict__ wgt,
                                                      const float* __restrict__ bias, const float* __restrict__ off,
                                                      const float* __restrict__ msk, float* __restrict__ out, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int wo = (int)(idx % s.Wo), ho = (int)((idx / s.Wo) % s.Ho);
  const int co = (int)((idx / ((size_t)s.Wo * s.Ho)) % s.Co), b = (int)(idx / ((size_t)s.Wo * s.Ho * s.Co));
  const int Cog = s.Co / s.group, Ck = s.C / s.group, cpdg = s.C / s.dg, K = s.kh * s.kw;
  const int g = co / Cog;
  const size_t HoWo = (size_t)s.Ho * s.Wo, HW = (size_t)s.H * s.W, po = (size_t)ho * s.Wo + wo;
  float acc = 0.f;
  for (int cl = 0; cl < Ck; ++cl) {
    const int ci = g * Ck + cl, d = ci / cpdg;
    const float* im = in + ((size_t)b * s.C + ci) * HW;
    const float* op = off + ((size_t)(b * s.dg + d) * 2 * K) * HoWo + po;
    const float* mp = msk + ((size_t)(b * s.dg + d) * K) * HoWo + po;
    const float* wp = wgt + ((size_t)co * Ck + cl) * K;
    for (int i = 0; i < s.kh; ++i)
      for (int j = 0; j < s.kw; ++j) {
        const int k = i * s.kw + j;
        const float hs = (float)(ho * s.sh - s.ph + i * s.dh) + op[(size_t)(2 * k) * HoWo];
        const float ws = (float)(wo * s.sw - s.pw + j * s.dw) + op[(size_t)(2 * k + 1) * HoWo];
        acc += (sample_plane(im, s.H, s.W, hs, ws) * mp[(size_t)k * HoWo]) * wp[k];
      }
  }
  out[idx] = bias[co] + acc;
}

// gradients w.r.t. offset, mask (modulated_deformable_col2im_coord_gpu_kernel, ...cuh:256-328) and input
// (modulated_deformable_col2im_gpu_kernel, ...cuh:196-254) in one pass: one lane per (b, deformable group, tap, ho, wo); the
// column gradient W^T . grad_out (modulated_deform_conv_cuda.cu:218-223) is formed on the fly.
__global__ void __launch_bounds__(256) dcn_bwd_data_kernel(DcnShape s, const float* __restrict__ in, const float* __restrict__ wgt,
                                                           const float* __restrict__ off, const float* __restrict__ msk,
                                                           const float* __restrict__ go, float* __restrict__ gin,
                                                           float* __restrict__ goff, float* __restrict__ gmsk, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int K = s.kh * s.kw;
  const int wo = (int)(idx % s.Wo), ho = (int)((idx / s.Wo) % s.Ho);
  const int k = (int)((idx / ((size_t)s.Wo * s.Ho)) % K);
  const int d = (int)((idx / ((size_t)s.Wo * s.Ho * K)) % s.dg), b = (int)(idx / ((size_t)s.Wo * s.Ho * K * s.dg));
  const int i = k / s.kw, j = k % s.kw;
  const int Cog = s.Co / s.group, Ck = s.C / s.group, cpdg = s.C / s.dg;
  const size_t HoWo = (size_t)s.Ho * s.Wo, HW = (size_t)s.H * s.W, po = (size_t)ho * s.Wo + wo;
  const float oh = off[((size_t)(b * s.dg + d) * 2 * K + 2 * k) * HoWo + po];
  const float ow = off[((size_t)(b * s.dg + d) * 2 * K + 2 * k + 1) * HoWo + po];
  const float m = msk[((size_t)(b * s.dg + d) * K + k) * HoWo + po];
  const float hs = (float)(ho * s.sh - s.ph + i * s.dh) + oh;
  const Tap t = make_tap(hs, (float)(wo * s.sw - s.pw + j * s.dw) + ow, s.H, s.W);
  // the reference's col2im launcher passes pad_h for pad_w (...cuh:372): the input gradient lands where THAT position falls
  const Tap ti = (s.ph == s.pw) ? t : make_tap(hs, (float)(wo * s.sw - s.ph + j * s.dw) + ow, s.H, s.W);
  float g_oh = 0.f, g_ow = 0.f, g_m = 0.f;
  for (int ci = d * cpdg; ci < (d + 1) * cpdg; ++ci) {
    const int g = ci / Ck, cl = ci - g * Ck;
    float gc = 0.f;
    for (int c2 = 0; c2 < Cog; ++c2) {
      const int co = g * Cog + c2;
      gc += wgt[((size_t)co * Ck + cl) * K + k] * go[((size_t)b * s.Co + co) * HoWo + po];
    }
    const float* im = in + ((size_t)b * s.C + ci) * HW;
    if (t.inside) {
      float v1, v2, v3, v4;
      corner_vals(im, s.H, s.W, t, v1, v2, v3, v4);
      g_m += gc * bilerp(t, v1, v2, v3, v4);                               // ...cuh:307
      // mdmcn_get_coordinate_weight (...cuh:83-125)
      const float a_w = (float)(t.wl + 1) - t.w, b_w = t.w - (float)t.wl;
      const float a_h = (float)(t.hl + 1) - t.h, b_h = t.h - (float)t.hl;
      const float dh = -a_w * v1 - b_w * v2 + a_w * v3 + b_w * v4;         // bp_dir == 0
      const float dw = -a_h * v1 + a_h * v2 - b_h * v3 + b_h * v4;         // bp_dir == 1
      g_oh += dh * gc * m;                                                 // ...cuh:312
      g_ow += dw * gc * m;
    }
    if (gin != nullptr && ti.inside) {
      const float top = gc * m;                                            // ...cuh:232
      float* gp = gin + ((size_t)b * s.C + ci) * HW;
      const int h0 = ti.hl, w0 = ti.wl, h1 = h0 + 1, w1 = w0 + 1;
      // mdmcn_get_gradient_weight (...cuh:56-81) for the in-image corners
      const float wh0 = (float)(h0 + 1) - ti.h, wh1 = (ti.h + 1.f) - (float)h1;
      const float ww0 = (float)(w0 + 1) - ti.w, ww1 = (ti.w + 1.f) - (float)w1;
      if (h0 >= 0 && w0 >= 0) atomicAdd(gp + (size_t)h0 * s.W + w0, wh0 * ww0 * top);
      if (h0 >= 0 && w1 <= s.W - 1) atomicAdd(gp + (size_t)h0 * s.W + w1, wh0 * ww1 * top);
      if (h1 <= s.H - 1 && w0 >= 0) atomicAdd(gp + (size_t)h1 * s.W + w0, wh1 * ww0 * top);
      if (h1 <= s.H - 1 && w1 <= s.W - 1) atomicAdd(gp + (size_t)h1 * s.W + w1, wh1 * ww1 * top);
    }
  }
  if (goff != nullptr) {
    goff[((size_t)(b * s.dg + d) * 2 * K + 2 * k) * HoWo + po] = g_oh;
    goff[((size_t)(b * s.dg + d) * 2 * K + 2 * k + 1) * HoWo + po] = g_ow;
  }
  if (gmsk != nullptr) gmsk[((size_t)(b * s.dg + d) * K + k) * HoWo + po] = g_m;
}

__device__ __forceinline__ float block_sum_256(float v) {
  __shared__ float part[4];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  return part[0] + part[1] + part[2] + part[3];
}

// grad_weight[co][cl][k] = sum over (b, ho, wo) of grad_out * column (modulated_deform_conv_cuda.cu:249-272); blockIdx.x = weight
// element, blockIdx.y = slab of the B*Ho*Wo positions; partial sums meet in one fp32 atomic per block.
__global__ void __launch_bounds__(256) dcn_bwd_weight_kernel(DcnShape s, const float* __restrict__ in, const float* __restrict__ off,
                                                             const float* __restrict__ msk, const float* __restrict__ go,
                                                             float* __restrict__ gw, size_t npos, size_t slab) {
  const int K = s.kh * s.kw, Cog = s.Co / s.group, Ck = s.C / s.group, cpdg = s.C / s.dg;
  const int k = blockIdx.x % K, cl = (blockIdx.x / K) % Ck, co = blockIdx.x / (K * Ck);
  const int g = co / Cog, ci = g * Ck + cl, d = ci / cpdg;
  const int i = k / s.kw, j = k % s.kw;
  const size_t HoWo = (size_t)s.Ho * s.Wo, HW = (size_t)s.H * s.W;
  const size_t n0 = (size_t)blockIdx.y * slab, n1 = n0 + slab < npos ? n0 + slab : npos;
  float acc = 0.f;
  for (size_t n = n0 + threadIdx.x; n < n1; n += 256) {
    const int b = (int)(n / HoWo);
    const size_t po = n - (size_t)b * HoWo;
    const int ho = (int)(po / s.Wo), wo = (int)(po - (size_t)ho * s.Wo);
    const float oh = off[((size_t)(b * s.dg + d) * 2 * K + 2 * k) * HoWo + po];
    const float ow = off[((size_t)(b * s.dg + d) * 2 * K + 2 * k + 1) * HoWo + po];
    const float m = msk[((size_t)(b * s.dg + d) * K + k) * HoWo + po];
    const float val = sample_plane(in + ((size_t)b * s.C + ci) * HW, s.H, s.W, (float)(ho * s.sh - s.ph + i * s.dh) + oh,
                                   (float)(wo * s.sw - s.pw + j * s.dw) + ow);
    acc += go[((size_t)b * s.Co + co) * HoWo + po] * (val * m);
  }
  acc = block_sum_256(acc);
  if (threadIdx.x == 0) atomicAdd(gw + blockIdx.x, acc);
}

// grad_bias[co] = sum over (b, ho, wo) of grad_out (at::addmv with ones, modulated_deform_conv_cuda.cu:273)
__global__ void __launch_bounds__(256) dcn_bwd_bias_kernel(DcnShape s, const float* __restrict__ go, float* __restrict__ gb,
                                                           size_t npos, size_t slab) {
  const int co = blockIdx.x;
  const size_t HoWo = (size_t)s.Ho * s.Wo;
  const size_t n0 = (size_t)blockIdx.y * slab, n1 = n0 + slab < npos ? n0 + slab : npos;
  float acc = 0.f;
  for (size_t n = n0 + threadIdx.x; n < n1; n += 256) {
    const size_t b = n / HoWo, po = n - b * HoWo;
    acc += go[(b * s.Co + co) * HoWo + po];
  }
  acc = block_sum_256(acc);
  if (threadIdx.x == 0) atomicAdd(gb + co, acc);
}

int check_dcn_args(DcnShape& s, int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                   int group, int dg, int im2col_step) {
  if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cout <= 0 || kh <= 0 || kw <= 0 || sh <= 0 || sw <= 0 || dh <= 0 || dw <= 0 || ph < 0 ||
      pw < 0 || group <= 0 || dg <= 0 || im2col_step <= 0)
    return dcn_fail(DD_ERR_INVALID_ARG, "sizes, strides, dilations, groups and im2col_step must be positive (paddings non-negative)");
  const int step = B < im2col_step ? B : im2col_step;
  if (B % step != 0)                                           // modulated_deform_conv_cuda.cu:58
    return dcn_fail(DD_ERR_INVALID_ARG, "batch(%d) must divide im2col_step(%d)", B, step);
  if (C % group != 0 || Cout % group != 0)                     // :60-61
    return dcn_fail(DD_ERR_INVALID_ARG, "channels(%d) and channels_out(%d) must divide group(%d)", C, Cout, group);
  if (C % dg != 0) return dcn_fail(DD_ERR_INVALID_ARG, "channels(%d) must divide deformable_group(%d)", C, dg);
  const int Ho = (H + 2 * ph - (dh * (kh - 1) + 1)) / sh + 1, Wo = (W + 2 * pw - (dw * (kw - 1) + 1)) / sw + 1;   // :75-76
  if (Ho <= 0 || Wo <= 0 || H + 2 * ph < dh * (kh - 1) + 1 || W + 2 * pw < dw * (kw - 1) + 1)
    return dcn_fail(DD_ERR_INVALID_ARG, "empty output (%d x %d)", Ho, Wo);
  s = DcnShape{B, C, H, W, Cout, Ho, Wo, kh, kw, sh, sw, ph, pw, dh, dw, group, dg};
  return DD_OK;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Kernel choice for one propagation iteration.  DD_NLSPN_KERNEL (experiments; read per call) overrides the default: g4 | g1 = global
// gathers with 4 / 1 pixels per lane, l8 | l16 | l32 = LDS window with that many tile rows; suffix p = pair-load sampler (fallback),
// suffix q = the software-pipelined 3x3 LDS kernel.
struct PropChoice { int lds_th; int vec; bool pair; bool pipelined; };

template <int KF>
int launch_prop(const float* fin, const float* offset, const float* aff, const float* w, const float* b, float* out, const float* fix,
                float* blend, int B, int H, int W, hipStream_t st) {
  // 4 pixels per lane only for the 3x3 kernel: 5x5 / 7x7 rows would need 200+ VGPRs per lane at VEC = 4
  const bool can_vec4 = KF == 3 && (W % 4 == 0) && aligned16(fin) && aligned16(offset) && aligned16(aff) && aligned16(out) &&
                        (blend == nullptr || (aligned16(fix) && aligned16(blend)));
  const bool can_pair = W >= 2;
  // default (measured on MI355X, profiles/history/r01_run35_nlspn.md): LDS window, 16-row tiles, pair-load fallback
  PropChoice c{16, 1, can_pair, false};
  if (const char* e = getenv("DD_NLSPN_KERNEL")) {
    const std::string m(e);
    if (!m.empty() && m[0] == 'g') { c.lds_th = 0; c.vec = (m.size() > 1 && m[1] == '4' && can_vec4) ? 4 : 1; }
    if (!m.empty() && m[0] == 'l') { c.lds_th = atoi(m.c_str() + 1); if (c.lds_th != 8 && c.lds_th != 16 && c.lds_th != 32) c.lds_th = 16; }
    c.pair = can_pair && m.find('p') != std::string::npos;
    c.pipelined = m.find('q') != std::string::npos;
  }
#define DD_PROP_ARGS fin, offset, aff, w, b, out, fix, blend, H, W
  if (c.lds_th && c.pipelined && KF == 3) {
    const dim3 grid((W + 63) / 64, (H + c.lds_th - 1) / c.lds_th, B), block(256);
    if (c.pair) {
      if (c.lds_th == 8) hipLaunchKernelGGL((nlspn_prop_lds3_kernel<8, true>), grid, block, 0, st, DD_PROP_ARGS);
      else if (c.lds_th == 16) hipLaunchKernelGGL((nlspn_prop_lds3_kernel<16, true>), grid, block, 0, st, DD_PROP_ARGS);
      else hipLaunchKernelGGL((nlspn_prop_lds3_kernel<32, true>), grid, block, 0, st, DD_PROP_ARGS);
    } else {
      if (c.lds_th == 8) hipLaunchKernelGGL((nlspn_prop_lds3_kernel<8, false>), grid, block, 0, st, DD_PROP_ARGS);
      else if (c.lds_th == 16) hipLaunchKernelGGL((nlspn_prop_lds3_kernel<16, false>), grid, block, 0, st, DD_PROP_ARGS);
      else hipLaunchKernelGGL((nlspn_prop_lds3_kernel<32, false>), grid, block, 0, st, DD_PROP_ARGS);
    }
  } else if (c.lds_th) {
    const dim3 grid((W + 63) / 64, (H + c.lds_th - 1) / c.lds_th, B), block(256);
    if (c.pair) {
      if (c.lds_th == 8) hipLaunchKernelGGL((nlspn_prop_lds_kernel<KF, 8, true>), grid, block, 0, st, DD_PROP_ARGS);
      else if (c.lds_th == 16) hipLaunchKernelGGL((nlspn_prop_lds_kernel<KF, 16, true>), grid, block, 0, st, DD_PROP_ARGS);
      else hipLaunchKernelGGL((nlspn_prop_lds_kernel<KF, 32, true>), grid, block, 0, st, DD_PROP_ARGS);
    } else {
      if (c.lds_th == 8) hipLaunchKernelGGL((nlspn_prop_lds_kernel<KF, 8, false>), grid, block, 0, st, DD_PROP_ARGS);
      else if (c.lds_th == 16) hipLaunchKernelGGL((nlspn_prop_lds_kernel<KF, 16, false>), grid, block, 0, st, DD_PROP_ARGS);
      else hipLaunchKernelGGL((nlspn_prop_lds_kernel<KF, 32, false>), grid, block, 0, st, DD_PROP_ARGS);
    }
  } else {
    const int lanes_x = c.vec == 4 ? W / 4 : W;
    const dim3 grid((lanes_x + 63) / 64, (H + 3) / 4, B), block(64, 4);
    if constexpr (KF == 3) {
      if (c.vec == 4 && c.pair) hipLaunchKernelGGL((nlspn_prop_kernel<KF, 4, true>), grid, block, 0, st, DD_PROP_ARGS);
      if (c.vec == 4 && !c.pair) hipLaunchKernelGGL((nlspn_prop_kernel<KF, 4, false>), grid, block, 0, st, DD_PROP_ARGS);
    }
    if (c.vec != 4 && c.pair) hipLaunchKernelGGL((nlspn_prop_kernel<KF, 1, true>), grid, block, 0, st, DD_PROP_ARGS);
    if (c.vec != 4 && !c.pair) hipLaunchKernelGGL((nlspn_prop_kernel<KF, 1, false>), grid, block, 0, st, DD_PROP_ARGS);
  }
#undef DD_PROP_ARGS
  DCN_HIP(hipGetLastError());
  return DD_OK;
}

}  // namespace

extern "C" {

const char* dd_dcn_last_error(void) { return g_dcn_err.c_str(); }

int dd_dcn_forward(const float* input, const float* weight, const float* bias, const float* offset, const float* mask, float* output,
                   int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw, int group, int dg,
                   int im2col_step, void* stream) {
  if (!input || !weight || !bias || !offset || !mask || !output) return dcn_fail(DD_ERR_INVALID_ARG, "null tensor pointer");
  DcnShape s;
  if (int rc = check_dcn_args(s, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, group, dg, im2col_step)) return rc;
  const size_t total = (size_t)B * Cout * s.Ho * s.Wo;
  if (total > (size_t)INT_MAX * 256) return dcn_fail(DD_ERR_INVALID_ARG, "output too large");
  hipLaunchKernelGGL(dcn_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, s, input, weight, bias,
                     offset, mask, output, total);
  DCN_HIP(hipGetLastError());
  return DD_OK;
}

int dd_dcn_backward(const float* input, const float* weight, const float* bias, const float* offset, const float* mask,
                    const float* grad_output, float* grad_input, float* grad_offset, float* grad_mask, float* grad_weight,
                    float* grad_bias, int B, int C, int H, int W, int Cout, int kh, int kw, int sh, int sw, int ph, int pw, int dh, int dw,
                    int group, int dg, int im2col_step, void* stream) {
  (void)bias;
  if (!input || !weight || !offset || !mask || !grad_output) return dcn_fail(DD_ERR_INVALID_ARG, "null tensor pointer");
  DcnShape s;
  if (int rc = check_dcn_args(s, B, C, H, W, Cout, kh, kw, sh, sw, ph, pw, dh, dw, group, dg, im2col_step)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const int K = kh * kw;
  const size_t npos = (size_t)B * s.Ho * s.Wo;
  if (grad_input) DCN_HIP(hipMemsetAsync(grad_input, 0, (size_t)B * C * H * W * sizeof(float), st));
  if (grad_input || grad_offset || grad_mask) {
    const size_t total = npos * K * dg;
    if (total > (size_t)INT_MAX * 256) return dcn_fail(DD_ERR_INVALID_ARG, "problem too large");
    hipLaunchKernelGGL(dcn_bwd_data_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, s, input, weight, offset, mask,
                       grad_output, grad_input, grad_offset, grad_mask, total);
    DCN_HIP(hipGetLastError());
  }
  // slabs of positions so that a one-channel NLSPN-sized problem (9 weight elements) still fills the chip
  size_t nslab = (npos + 8191) / 8192;
  if (nslab > 1024) nslab = 1024;
  const size_t slab = (npos + nslab - 1) / nslab;
  if (grad_weight) {
    const size_t nw = (size_t)Cout * (C / group) * K;
    DCN_HIP(hipMemsetAsync(grad_weight, 0, nw * sizeof(float), st));
    hipLaunchKernelGGL(dcn_bwd_weight_kernel, dim3((unsigned)nw, (unsigned)nslab), dim3(256), 0, st, s, input, offset, mask, grad_output,
                       grad_weight, npos, slab);
    DCN_HIP(hipGetLastError());
  }
  if (grad_bias) {
    DCN_HIP(hipMemsetAsync(grad_bias, 0, (size_t)Cout * sizeof(float), st));
    hipLaunchKernelGGL(dcn_bwd_bias_kernel, dim3((unsigned)Cout, (unsigned)nslab), dim3(256), 0, st, s, grad_output, grad_bias, npos, slab);
    DCN_HIP(hipGetLastError());
  }
  return DD_OK;
}

int dd_nlspn_offset_affinity(const float* offset_aff, const float* confidence, const float* aff_scale_const, const float* w_conf,
                             const float* b_conf, float* offset, float* aff, int B, int H, int W, int k_f, int affinity, int conf_prop,
                             int legacy, void* stream) {
  if (!offset_aff || !aff_scale_const || !offset || !aff) return dcn_fail(DD_ERR_INVALID_ARG, "null tensor pointer");
  if (B <= 0 || H <= 0 || W <= 0) return dcn_fail(DD_ERR_INVALID_ARG, "B, H, W must be positive");
  if (affinity < DD_AFF_AS || affinity > DD_AFF_TGASS) return dcn_fail(DD_ERR_INVALID_ARG, "unknown affinity mode %d", affinity);
  if (conf_prop && (!confidence || !w_conf || !b_conf))                     // `assert confidence is not None`, nlspnmodel.py:180
    return dcn_fail(DD_ERR_INVALID_ARG, "conf_prop needs confidence, w_conf and b_conf");
  const float* conf = conf_prop ? confidence : nullptr;
  const dim3 grid((W + 63) / 64, (H + 3) / 4, B), block(64, 4);
  hipStream_t st = (hipStream_t)stream;
  switch (k_f) {
    case 3: hipLaunchKernelGGL(nlspn_affinity_kernel<3>, grid, block, 0, st, offset_aff, conf, aff_scale_const, w_conf, b_conf, offset, aff, H, W, affinity, legacy); break;
    case 5: hipLaunchKernelGGL(nlspn_affinity_kernel<5>, grid, block, 0, st, offset_aff, conf, aff_scale_const, w_conf, b_conf, offset, aff, H, W, affinity, legacy); break;
    case 7: hipLaunchKernelGGL(nlspn_affinity_kernel<7>, grid, block, 0, st, offset_aff, conf, aff_scale_const, w_conf, b_conf, offset, aff, H, W, affinity, legacy); break;
    default: return dcn_fail(DD_ERR_UNSUPPORTED, "prop_kernel %d: this build has 3, 5 and 7 (the reference asserts an odd size, nlspnmodel.py:35)", k_f);
  }
  DCN_HIP(hipGetLastError());
  return DD_OK;
}

int dd_nlspn_guided_offset_affinity(const float* guidance, const float* conv_weight, const float* conv_bias, const float* confidence,
                                    const float* aff_scale_const, const float* w_conf, const float* b_conf, float* offset, float* aff,
                                    int B, int ch_g, int H, int W, int k_g, int k_f, int affinity, int conf_prop, int legacy, void* stream) {
  if (!guidance || !conv_weight || !conv_bias || !aff_scale_const || !offset || !aff) return dcn_fail(DD_ERR_INVALID_ARG, "null tensor pointer");
  if (B <= 0 || H <= 0 || W <= 0) return dcn_fail(DD_ERR_INVALID_ARG, "B, H, W must be positive");
  if (affinity < DD_AFF_AS || affinity > DD_AFF_TGASS) return dcn_fail(DD_ERR_INVALID_ARG, "unknown affinity mode %d", affinity);
  if (conf_prop && (!confidence || !w_conf || !b_conf)) return dcn_fail(DD_ERR_INVALID_ARG, "conf_prop needs confidence, w_conf and b_conf");
  if (ch_g != 8 || k_g != 3 || k_f != 3)
    return dcn_fail(DD_ERR_UNSUPPORTED, "fused guidance convolution is built for ch_g 8, k_g 3, k_f 3 (NLSPNModel's geometry); got %d, %d, %d: "
                                        "run the convolution separately and call dd_nlspn_offset_affinity", ch_g, k_g, k_f);
  const dim3 grid((W + 63) / 64, (H + 7) / 8, B), block(256);
  hipLaunchKernelGGL(nlspn_guide_affinity_kernel<8>, grid, block, 0, (hipStream_t)stream, guidance, conv_weight, conv_bias,
                     conf_prop ? confidence : nullptr, aff_scale_const, w_conf, b_conf, offset, aff, H, W, affinity, legacy);
  DCN_HIP(hipGetLastError());
  return DD_OK;
}

int dd_nlspn_workspace_bytes(int B, int H, int W, int preserve_input, int64_t* bytes) {
  if (!bytes || B <= 0 || H <= 0 || W <= 0) return dcn_fail(DD_ERR_INVALID_ARG, "bad arguments");
  *bytes = preserve_input ? (int64_t)2 * B * H * W * (int64_t)sizeof(float) : 0;
  return DD_OK;
}

int dd_nlspn_propagate(const float* feat_init, const float* offset, const float* aff, const float* feat_fix, const float* w,
                       const float* b, float* feat_list, void* workspace, int B, int H, int W, int k_f, int prop_time,
                       int preserve_input, void* stream) {
  if (!feat_init || !offset || !aff || !w || !b || !feat_list) return dcn_fail(DD_ERR_INVALID_ARG, "null tensor pointer");
  if (B <= 0 || H <= 0 || W <= 0 || prop_time <= 0) return dcn_fail(DD_ERR_INVALID_ARG, "B, H, W, prop_time must be positive");
  if (preserve_input && (!feat_fix || !workspace))                          // `assert feat_init.shape == feat_fix.shape`, nlspnmodel.py:188
    return dcn_fail(DD_ERR_INVALID_ARG, "preserve_input needs feat_fix and a workspace of dd_nlspn_workspace_bytes()");
  if (k_f != 3 && k_f != 5 && k_f != 7)
    return dcn_fail(DD_ERR_UNSUPPORTED, "prop_kernel %d: this build has 3, 5 and 7", k_f);
  hipStream_t st = (hipStream_t)stream;
  const size_t n = (size_t)B * H * W;
  float* ws[2] = {nullptr, nullptr};
  if (preserve_input) {
    ws[0] = static_cast<float*>(workspace);
    ws[1] = ws[0] + n;
    hipLaunchKernelGGL(nlspn_blend_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, feat_init, feat_fix, ws[0], n);
    DCN_HIP(hipGetLastError());
  }
  for (int it = 0; it < prop_time; ++it) {
    const float* fin = preserve_input ? ws[it & 1] : (it == 0 ? feat_init : feat_list + (size_t)(it - 1) * n);
    float* out = feat_list + (size_t)it * n;
    float* blend = (preserve_input && it + 1 < prop_time) ? ws[(it + 1) & 1] : nullptr;
    int rc;
    switch (k_f) {
      case 3: rc = launch_prop<3>(fin, offset, aff, w, b, out, feat_fix, blend, B, H, W, st); break;
      case 5: rc = launch_prop<5>(fin, offset, aff, w, b, out, feat_fix, blend, B, H, W, st); break;
      default: rc = launch_prop<7>(fin, offset, aff, w, b, out, feat_fix, blend, B, H, W, st); break;
    }
    if (rc) return rc;
  }
  return DD_OK;
}

}  // extern "C"
