// dd_misc.hip -- layout conversion, DDIM elementwise tails, q_sample and the latent codec.
//
// All of these are HBM-bound streaming kernels that run once per image (or once per call), not once
// per DDIM step; they are written for coalescing (consecutive lanes <-> consecutive addresses on the
// wide side of every transfer), not for the matrix cores.
#include "dd_elem.h"
#include "dd_gcn.h"

namespace dd {
// torch.relu / torch.clamp keep a NaN; fmaxf (v_max_f32) returns the other operand.  The once-per-image kernels of this file follow torch.
__device__ __forceinline__ float relu_keep_nan(float v) { return (v < 0.f) ? 0.f : v; }


__device__ __forceinline__ uint32_t cvt_bf16(float f) {
  uint32_t u = __builtin_bit_cast(uint32_t, f);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return u >> 16;
}
__device__ __forceinline__ uint32_t cvt_f16(float f) { return (uint32_t)__builtin_bit_cast(uint16_t, (_Float16)f); }
__device__ __forceinline__ float ld_elem(const void* base, size_t idx, int ek) {
  if (ek == EK_F32) return reinterpret_cast<const float*>(base)[idx];
  const uint32_t u = reinterpret_cast<const uint16_t*>(base)[idx];
  return ek == EK_BF16 ? __builtin_bit_cast(float, u << 16) : (float)__builtin_bit_cast(_Float16, (uint16_t)u);
}

// ------------------------------------------------------------------------------------------------
// NCHW fp32 -> NHWC (fp32 / bf16 / f16).  A 256-thread block transposes a [64 px][64 ch] tile through
// LDS: reads are 256-B runs along pixels of one channel plane, writes are 64 channels of a pixel.
// ------------------------------------------------------------------------------------------------
template <int EK>
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ src, void* __restrict__ dst,
                                                           int C, int Cd, long long HW, int blocked) {
  // C = channels of the NCHW source, Cd >= C = channels of the destination (zero-filled beyond C: pyramid widths that are not a
  // multiple of the 32-channel block, e.g. MPViT's 216 -> 224)
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  const long long p0 = (long long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
  const int px = tid & 63, cs = tid >> 6;
  const float* s = src + ((size_t)b * C) * HW;
  if ((HW & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
    // 16-B loads along the pixels of a channel plane: four independent loads per thread in flight (the scalar form below has 16 x 4 B)
    const int q = tid & 15, crow = tid >> 4;          // 16 quads of pixels x 16 channels per pass, 4 passes
    float4 v4[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int c = cc * 16 + crow;
      v4[cc] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p0 + 4 * q < HW && c0 + c < C) v4[cc] = *reinterpret_cast<const float4*>(s + (size_t)(c0 + c) * HW + p0 + 4 * q);   // HW % 4 == 0: a quad is inside or outside
    }
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int c = cc * 16 + crow;
      tile[4 * q + 0][c] = v4[cc].x; tile[4 * q + 1][c] = v4[cc].y; tile[4 * q + 2][c] = v4[cc].z; tile[4 * q + 3][c] = v4[cc].w;
    }
  } else {
#pragma unroll 4
  for (int cc = 0; cc < 16; ++cc) {
    const int c = cc * 4 + cs;
    float v = 0.f;
    if (p0 + px < HW && c0 + c < C) v = s[(size_t)(c0 + c) * HW + p0 + px];
    tile[px][c] = v;
  }
  }
  __syncthreads();
  const int opx = tid >> 2, part = tid & 3;           // 4 threads per pixel, 16 channels each
  if (p0 + opx >= HW) return;
  const int cbase = c0 + part * 16;
  if (cbase >= Cd) return;                             // Cd is a multiple of 16 for every tensor we convert
  // destination offset in the activation layout of dd_elem.h (channel-blocked when Cd >= 32); h*w is all that matters
  const size_t o = (Cd < ACT_CB || !blocked) ? ((size_t)b * HW + p0 + opx) * Cd + cbase
                                : (((size_t)b * (Cd / ACT_CB) + cbase / ACT_CB) * HW + p0 + opx) * ACT_CB + (cbase % ACT_CB);
  if constexpr (EK == EK_F32) {
    float4* d = reinterpret_cast<float4*>(reinterpret_cast<float*>(dst) + o);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      d[i] = make_float4(tile[opx][part * 16 + 4 * i], tile[opx][part * 16 + 4 * i + 1],
                         tile[opx][part * 16 + 4 * i + 2], tile[opx][part * 16 + 4 * i + 3]);
  } else {
    uint32_t wv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float lo = tile[opx][part * 16 + 2 * i], hi = tile[opx][part * 16 + 2 * i + 1];
      wv[i] = (EK == EK_BF16) ? (cvt_bf16(lo) | (cvt_bf16(hi) << 16)) : (cvt_f16(lo) | (cvt_f16(hi) << 16));
    }
    uint4* d = reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(dst) + o);
    d[0] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
    d[1] = make_uint4(wv[4], wv[5], wv[6], wv[7]);
  }
}

hipError_t launch_nchw_to_nhwc_padded(const float* src, void* dst, int ek, int B, int C, int Cd, int h, int w, int blocked, hipStream_t s) {
  if (Cd % 16 != 0 || Cd < C || C <= 0) return hipErrorInvalidValue;
  const long long HW = (long long)h * w;
  dim3 grid((unsigned)((HW + 63) / 64), (unsigned)((Cd + 63) / 64), (unsigned)B);
  if (ek == EK_F32) hipLaunchKernelGGL(nchw_to_nhwc_kernel<EK_F32>, grid, dim3(256), 0, s, src, dst, C, Cd, HW, blocked);
  else if (ek == EK_BF16) hipLaunchKernelGGL(nchw_to_nhwc_kernel<EK_BF16>, grid, dim3(256), 0, s, src, dst, C, Cd, HW, blocked);
  else hipLaunchKernelGGL(nchw_to_nhwc_kernel<EK_F16>, grid, dim3(256), 0, s, src, dst, C, Cd, HW, blocked);
  return hipGetLastError();
}
hipError_t launch_nchw_to_nhwc(const float* src, void* dst, int ek, int B, int C, int h, int w, int blocked, hipStream_t s) {
  return launch_nchw_to_nhwc_padded(src, dst, ek, B, C, C, h, w, blocked, s);
}

// ------------------------------------------------------------------------------------------------
// Swin variant: bilinear upsample (align_corners=True) NCHW fp32 (ch,cw) -> channel-blocked (h,w).  Same tiling as
// nchw_to_nhwc_kernel: a block produces [64 output pixels][64 channels]; the 4 source taps of a pixel are read
// along pixels of one channel plane (neighbouring lanes hit neighbouring / identical source columns).
// torch semantics: src = dst * (in-1)/(out-1) (scale computed in fp32), i0 = floor, lambda1 = src - i0.
// ------------------------------------------------------------------------------------------------
template <int EK>
__global__ void __launch_bounds__(256) upsample_to_blocked_kernel(const float* __restrict__ src, void* __restrict__ dst, int C,
                                                                  int ch, int cw, int h, int w) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  const long long HW = (long long)h * w, SHW = (long long)ch * cw;
  const long long p0 = (long long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
  const int px = tid & 63, cs = tid >> 6;
  const long long p = p0 + px;
  const bool pin = p < HW;
  const int oy = pin ? (int)(p / w) : 0, ox = pin ? (int)(p - (long long)oy * w) : 0;
  const float sy = (h > 1) ? (float)(ch - 1) / (float)(h - 1) : 0.f;
  const float sx = (w > 1) ? (float)(cw - 1) / (float)(w - 1) : 0.f;
  const float fy = sy * (float)oy, fx = sx * (float)ox;
  const int y0 = min((int)fy, ch - 1), x0 = min((int)fx, cw - 1);
  const int y1 = min(y0 + 1, ch - 1), x1 = min(x0 + 1, cw - 1);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const float* sb = src + (size_t)b * C * SHW;
#pragma unroll 4
  for (int cc = 0; cc < 16; ++cc) {
    const int c = cc * 4 + cs;
    float v = 0.f;
    if (pin && c0 + c < C) {
      const float* pl = sb + (size_t)(c0 + c) * SHW;
      v = hy * (hx * pl[(size_t)y0 * cw + x0] + lx * pl[(size_t)y0 * cw + x1]) +
          ly * (hx * pl[(size_t)y1 * cw + x0] + lx * pl[(size_t)y1 * cw + x1]);
    }
    tile[px][c] = v;
  }
  __syncthreads();
  const int opx = tid >> 2, part = tid & 3;
  if (p0 + opx >= HW) return;
  const int cbase = c0 + part * 16;
  if (cbase >= C) return;
  const size_t o = (((size_t)b * (C / ACT_CB) + cbase / ACT_CB) * HW + p0 + opx) * ACT_CB + (cbase % ACT_CB);
  if constexpr (EK == EK_F32) {
    float4* d = reinterpret_cast<float4*>(reinterpret_cast<float*>(dst) + o);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      d[i] = make_float4(tile[opx][part * 16 + 4 * i], tile[opx][part * 16 + 4 * i + 1],
                         tile[opx][part * 16 + 4 * i + 2], tile[opx][part * 16 + 4 * i + 3]);
  } else {
    uint32_t wv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float lo = tile[opx][part * 16 + 2 * i], hi = tile[opx][part * 16 + 2 * i + 1];
      wv[i] = (EK == EK_BF16) ? (cvt_bf16(lo) | (cvt_bf16(hi) << 16)) : (cvt_f16(lo) | (cvt_f16(hi) << 16));
    }
    uint4* d = reinterpret_cast<uint4*>(reinterpret_cast<uint16_t*>(dst) + o);
    d[0] = make_uint4(wv[0], wv[1], wv[2], wv[3]);
    d[1] = make_uint4(wv[4], wv[5], wv[6], wv[7]);
  }
}
hipError_t launch_upsample_to_blocked(const float* src, void* dst, int ek, int B, int C, int ch, int cw, int h, int w, hipStream_t s) {
  if (C % 64 != 0) return hipErrorInvalidValue;
  const long long HW = (long long)h * w;
  dim3 grid((unsigned)((HW + 63) / 64), (unsigned)(C / 64), (unsigned)B);
  if (ek == EK_F32) hipLaunchKernelGGL(upsample_to_blocked_kernel<EK_F32>, grid, dim3(256), 0, s, src, dst, C, ch, cw, h, w);
  else if (ek == EK_BF16) hipLaunchKernelGGL(upsample_to_blocked_kernel<EK_BF16>, grid, dim3(256), 0, s, src, dst, C, ch, cw, h, w);
  else hipLaunchKernelGGL(upsample_to_blocked_kernel<EK_F16>, grid, dim3(256), 0, s, src, dst, C, ch, cw, h, w);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Swin variant, condition map already in the activation layout (written by dd_condition at stride 4): bilinear upsample
// (align_corners=True, same fp32 formula as upsample_to_blocked_kernel) blocked (ch,cw) -> blocked (h,w).
// One thread per (output pixel, 16-B piece): 4 taps of 16 B each.
// ------------------------------------------------------------------------------------------------
template <int EK>
__global__ void __launch_bounds__(256) upsample_blocked_kernel(const void* __restrict__ src, void* __restrict__ dst, int nblk,
                                                              int ch, int cw, int h, int w, long long total) {
  constexpr int EPP = (EK == EK_F32) ? 4 : 8;
  constexpr int PPB = ACT_CB / EPP;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int piece = (int)(i % PPB);
  long long r = i / PPB;
  const int ox = (int)(r % w); r /= w;
  const int oy = (int)(r % h); r /= h;                  // r = b * nblk + channel block
  const float sy = (h > 1) ? (float)(ch - 1) / (float)(h - 1) : 0.f;
  const float sx = (w > 1) ? (float)(cw - 1) / (float)(w - 1) : 0.f;
  const float fy = sy * (float)oy, fx = sx * (float)ox;
  const int y0 = min((int)fy, ch - 1), x0 = min((int)fx, cw - 1);
  const int y1 = min(y0 + 1, ch - 1), x1 = min(x0 + 1, cw - 1);
  const float ly = fy - (float)y0, lx = fx - (float)x0;
  const float hy = 1.f - ly, hx = 1.f - lx;
  const char* sb = reinterpret_cast<const char*>(src) + ((size_t)r * ch * cw) * ACT_CB * (16 / EPP) + piece * 16;
  auto tap = [&](int y, int x, float (&f)[EPP]) {
    const uint4 v = *reinterpret_cast<const uint4*>(sb + ((size_t)y * cw + x) * ACT_CB * (16 / EPP));
    const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
    if constexpr (EK == EK_F32) {
#pragma unroll
      for (int k = 0; k < 4; ++k) f[k] = __builtin_bit_cast(float, wv[k]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) { f[2 * k] = ld_elem(&wv[k], 0, EK); f[2 * k + 1] = ld_elem(&wv[k], 1, EK); }
    }
  };
  float a[EPP], b[EPP], c[EPP], d[EPP], o[EPP];
  tap(y0, x0, a); tap(y0, x1, b); tap(y1, x0, c); tap(y1, x1, d);
#pragma unroll
  for (int k = 0; k < EPP; ++k) o[k] = hy * (hx * a[k] + lx * b[k]) + ly * (hx * c[k] + lx * d[k]);
  char* db = reinterpret_cast<char*>(dst) + (((size_t)r * h + oy) * w + ox) * ACT_CB * (16 / EPP) + piece * 16;
  if constexpr (EK == EK_F32) {
    *reinterpret_cast<float4*>(db) = make_float4(o[0], o[1], o[2], o[3]);
  } else {
    uint32_t q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      q[k] = (EK == EK_BF16) ? (cvt_bf16(o[2 * k]) | (cvt_bf16(o[2 * k + 1]) << 16)) : (cvt_f16(o[2 * k]) | (cvt_f16(o[2 * k + 1]) << 16));
    *reinterpret_cast<uint4*>(db) = make_uint4(q[0], q[1], q[2], q[3]);
  }
}
hipError_t launch_upsample_blocked(const void* src, void* dst, int ek, int B, int C, int ch, int cw, int h, int w, hipStream_t s) {
  if (C % ACT_CB != 0) return hipErrorInvalidValue;
  const int nblk = C / ACT_CB;
  const int ppb = (ek == EK_F32) ? 8 : 4;
  const long long total = (long long)B * nblk * h * w * ppb;
  dim3 grid((unsigned)((total + 255) / 256));
  if (ek == EK_F32) hipLaunchKernelGGL(upsample_blocked_kernel<EK_F32>, grid, dim3(256), 0, s, src, dst, nblk, ch, cw, h, w, total);
  else if (ek == EK_BF16) hipLaunchKernelGGL(upsample_blocked_kernel<EK_BF16>, grid, dim3(256), 0, s, src, dst, nblk, ch, cw, h, w, total);
  else hipLaunchKernelGGL(upsample_blocked_kernel<EK_F16>, grid, dim3(256), 0, s, src, dst, nblk, ch, cw, h, w, total);
  return hipGetLastError();
}

// NHWC (any element kind) -> NCHW fp32; debug / small tensors only (one thread per output element).
__global__ void nhwc_to_nchw_kernel(const void* __restrict__ src, int ek, float* __restrict__ dst, int C, long long HW,
                                    long long total, int blocked) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long p = i % HW;
  const long long bc = i / HW;
  const int c = (int)(bc % C);
  const long long b = bc / C;
  const size_t o = (C < ACT_CB || !blocked) ? (size_t)(b * HW + p) * C + c
                                            : (((size_t)b * (C / ACT_CB) + c / ACT_CB) * HW + p) * ACT_CB + (c % ACT_CB);
  dst[i] = ld_elem(src, o, ek);
}
hipError_t launch_nhwc_to_nchw_f32(const void* src, int ek, float* dst, int B, int C, int h, int w, int blocked, hipStream_t s) {
  const long long HW = (long long)h * w, total = HW * C * B;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, ek, dst, C, HW, total, blocked);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Channel-blocked activations -> NCHW fp32 (the FPN's condition map handed back to the caller).  Mirror of
// nchw_to_nhwc_kernel: a block moves [64 px][64 ch] through LDS; reads are 16-B pieces of a pixel's channel block,
// writes are 256-B runs along pixels of one channel plane.
// ------------------------------------------------------------------------------------------------
template <int EK>
__global__ void __launch_bounds__(256) blocked_to_nchw_kernel(const void* __restrict__ src, float* __restrict__ dst, int C, long long HW, int accumulate) {
  __shared__ float tile[64][65];
  const int b = blockIdx.z;
  const long long p0 = (long long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
  const int ipx = tid >> 2, part = tid & 3;            // 4 threads per pixel, 16 channels each
  const int cbase = c0 + part * 16;
  if (p0 + ipx < HW && cbase < C) {
    const size_t o = (((size_t)b * (C / ACT_CB) + cbase / ACT_CB) * HW + p0 + ipx) * ACT_CB + (cbase % ACT_CB);
    if constexpr (EK == EK_F32) {
      const float4* sp = reinterpret_cast<const float4*>(reinterpret_cast<const float*>(src) + o);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 v = sp[i];
        tile[ipx][part * 16 + 4 * i] = v.x; tile[ipx][part * 16 + 4 * i + 1] = v.y;
        tile[ipx][part * 16 + 4 * i + 2] = v.z; tile[ipx][part * 16 + 4 * i + 3] = v.w;
      }
    } else {
      const uint4* sp = reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(src) + o);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const uint4 v = sp[i];
        const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          tile[ipx][part * 16 + 8 * i + 2 * j] = ld_elem(&wv[j], 0, EK);
          tile[ipx][part * 16 + 8 * i + 2 * j + 1] = ld_elem(&wv[j], 1, EK);
        }
      }
    }
  }
  __syncthreads();
  const int px = tid & 63, cs = tid >> 6;
  if (p0 + px >= HW) return;
  float* d = dst + ((size_t)b * C) * HW;
#pragma unroll 4
  for (int cc = 0; cc < 16; ++cc) {
    const int c = cc * 4 + cs;
    if (c0 + c < C) {
      float* o = d + (size_t)(c0 + c) * HW + p0 + px;
      *o = accumulate ? *o + tile[px][c] : tile[px][c];
    }
  }
}
hipError_t launch_blocked_to_nchw(const void* src, int ek, float* dst, int B, int C, int h, int w, int accumulate, hipStream_t s) {
  if (C % ACT_CB != 0) return hipErrorInvalidValue;
  const long long HW = (long long)h * w;
  dim3 grid((unsigned)((HW + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)B);
  if (ek == EK_F32) hipLaunchKernelGGL(blocked_to_nchw_kernel<EK_F32>, grid, dim3(256), 0, s, src, dst, C, HW, accumulate);
  else if (ek == EK_BF16) hipLaunchKernelGGL(blocked_to_nchw_kernel<EK_BF16>, grid, dim3(256), 0, s, src, dst, C, HW, accumulate);
  else hipLaunchKernelGGL(blocked_to_nchw_kernel<EK_F16>, grid, dim3(256), 0, s, src, dst, C, HW, accumulate);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// adaptive_avg_pool2d on channel-blocked activations (reference ...res.py:116: the 2x-upsampled top-down term is
// pooled to the lateral map's size when the backbone sizes are odd).  torch semantics: output (oy, ox) averages input
// rows floor(oy*IH/OH) .. ceil((oy+1)*IH/OH)-1 (same for columns), fp32 accumulation in row-major order.
// One thread per (output pixel, 16-B piece): consecutive lanes <-> consecutive pieces of consecutive pixels.
// ------------------------------------------------------------------------------------------------
template <int EK>
__global__ void __launch_bounds__(256) adaptive_pool_blocked_kernel(const void* __restrict__ src, void* __restrict__ dst,
                                                                   int nblk, int ih, int iw, int oh, int ow, long long total) {
  constexpr int EPP = (EK == EK_F32) ? 4 : 8;           // elements per 16-B piece
  constexpr int PPB = ACT_CB / EPP;                     // pieces per pixel of one channel block
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int piece = (int)(i % PPB);
  long long r = i / PPB;
  const int ox = (int)(r % ow); r /= ow;
  const int oy = (int)(r % oh); r /= oh;                // r = b * nblk + channel block
  const int ys = (int)(((long long)oy * ih) / oh), ye = (int)((((long long)oy + 1) * ih + oh - 1) / oh);
  const int xs = (int)(((long long)ox * iw) / ow), xe = (int)((((long long)ox + 1) * iw + ow - 1) / ow);
  float acc[EPP];
#pragma unroll
  for (int k = 0; k < EPP; ++k) acc[k] = 0.f;
  const char* sb = reinterpret_cast<const char*>(src) + ((size_t)r * ih * iw) * ACT_CB * (16 / EPP) + piece * 16;
  for (int y = ys; y < ye; ++y)
    for (int x = xs; x < xe; ++x) {
      const uint4 v = *reinterpret_cast<const uint4*>(sb + ((size_t)y * iw + x) * ACT_CB * (16 / EPP));
      const uint32_t wv[4] = {v.x, v.y, v.z, v.w};
      if constexpr (EK == EK_F32) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] += __builtin_bit_cast(float, wv[k]);
      } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc[2 * k] += ld_elem(&wv[k], 0, EK); acc[2 * k + 1] += ld_elem(&wv[k], 1, EK); }
      }
    }
  const float inv = (float)((ye - ys) * (xe - xs));
#pragma unroll
  for (int k = 0; k < EPP; ++k) acc[k] /= inv;
  char* db = reinterpret_cast<char*>(dst) + (((size_t)r * oh + oy) * ow + ox) * ACT_CB * (16 / EPP) + piece * 16;
  if constexpr (EK == EK_F32) {
    *reinterpret_cast<float4*>(db) = make_float4(acc[0], acc[1], acc[2], acc[3]);
  } else {
    uint32_t o[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      o[k] = (EK == EK_BF16) ? (cvt_bf16(acc[2 * k]) | (cvt_bf16(acc[2 * k + 1]) << 16)) : (cvt_f16(acc[2 * k]) | (cvt_f16(acc[2 * k + 1]) << 16));
    *reinterpret_cast<uint4*>(db) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}
hipError_t launch_adaptive_pool_blocked(const void* src, void* dst, int ek, int B, int C, int ih, int iw, int oh, int ow, hipStream_t s) {
  if (C % ACT_CB != 0) return hipErrorInvalidValue;
  const int nblk = C / ACT_CB;
  const int ppb = (ek == EK_F32) ? 8 : 4;
  const long long total = (long long)B * nblk * oh * ow * ppb;
  dim3 grid((unsigned)((total + 255) / 256));
  if (ek == EK_F32) hipLaunchKernelGGL(adaptive_pool_blocked_kernel<EK_F32>, grid, dim3(256), 0, s, src, dst, nblk, ih, iw, oh, ow, total);
  else if (ek == EK_BF16) hipLaunchKernelGGL(adaptive_pool_blocked_kernel<EK_BF16>, grid, dim3(256), 0, s, src, dst, nblk, ih, iw, oh, ow, total);
  else hipLaunchKernelGGL(adaptive_pool_blocked_kernel<EK_F16>, grid, dim3(256), 0, s, src, dst, nblk, ih, iw, oh, ow, total);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Tail of the loop: x_0 = c1*x + c2*relu(gn4(y4))  (mode 0)   or   eps = relu(gn4(y4))  (mode 1),
// NHWC fp32 in, NCHW fp32 out.  One thread per pixel: 64-B reads per lane, 16 plane-coalesced writes.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) final_kernel(const float* __restrict__ x, const float* __restrict__ y4,
                                                    const double* __restrict__ stats, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, const float* __restrict__ c1c2,
                                                    int step, int mode, float* __restrict__ out, long long HW) {
  __shared__ double s_sum[8];
  __shared__ float s_a[LATENT_C], s_b[LATENT_C];
  __shared__ float s_poison;      // 0, or NaN when y4's statistics are not finite: the image is poisoned (dd_igemm2.hip) and comes out as NaN, as from the reference
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  if (tid < 8) {
    const double* st = stats + (size_t)b * STAT_SLOTS * STAT_STRIDE;
    double acc = 0.0;
    for (int s = 0; s < STAT_SLOTS; ++s) acc += st[s * STAT_STRIDE + tid];
    s_sum[tid] = acc;
  }
  __syncthreads();
  if (tid < LATENT_C) {
    const int grp = tid / (LATENT_C / GN_GROUPS);
    const double cnt = (double)HW * (LATENT_C / GN_GROUPS);
    const double mean = s_sum[grp * 2] / cnt;
    double var = s_sum[grp * 2 + 1] / cnt - mean * mean;
    var = var > 0.0 ? var : 0.0;
    const double a = (double)gamma[tid] / sqrt(var + (double)GN_EPS);
    s_a[tid] = (float)a;
    s_b[tid] = (float)((double)beta[tid] - mean * a);
  }
  if (tid == 0) {
    double z = 0.0;
    for (int k = 0; k < 8; ++k) z += s_sum[k] - s_sum[k];
    s_poison = (z == 0.0) ? 0.f : __builtin_nanf("");
  }
  __syncthreads();
  const long long p = (long long)blockIdx.x * blockDim.x + tid;
  if (p >= HW) return;
  const float poison = s_poison;
  float c1 = 0.f, c2 = 1.f;
  if (mode == 0) { c1 = c1c2[2 * step]; c2 = c1c2[2 * step + 1]; }
  const float4* yv = reinterpret_cast<const float4*>(y4 + ((size_t)b * HW + p) * LATENT_C);
  const float4* xv = reinterpret_cast<const float4*>(x + ((size_t)b * HW + p) * LATENT_C);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 yy = yv[q];
    float4 xx = make_float4(0.f, 0.f, 0.f, 0.f);
    if (mode == 0) xx = xv[q];
    const float ys[4] = {yy.x, yy.y, yy.z, yy.w};
    const float xs[4] = {xx.x, xx.y, xx.z, xx.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = q * 4 + i;
      const float e = fmaxf(fmaf(s_a[c], ys[i], s_b[c]), 0.f);
      out[((size_t)b * LATENT_C + c) * HW + p] = ((mode == 0) ? (c1 * xs[i] + c2 * e) : e) + poison;
    }
  }
}
hipError_t launch_final(const float* x, const float* y4, const double* stats, const float* gamma, const float* beta,
                        const float* c1c2, int step, int mode, float* out_nchw, int B, int h, int w, hipStream_t s) {
  const long long HW = (long long)h * w;
  dim3 grid((unsigned)((HW + 255) / 256), (unsigned)B);
  hipLaunchKernelGGL(final_kernel, grid, dim3(256), 0, s, x, y4, stats, gamma, beta, c1c2, step, mode, out_nchw, HW);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Time-embedding term of conv3 hoisted out of the DDIM loop (runs once per weight commit): the embedding E[t] is
// spatially constant, so conv3(E[t]) at a pixel is the sum over the taps that fall inside the image of W3_tap . E[t].
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64) etab_kernel(const float* __restrict__ w3, const float* __restrict__ emb, float* __restrict__ etab) {
  const int t = blockIdx.x, co = threadIdx.x;
  const float* e = emb + (size_t)t * COND_C;
  double tot = 0.0;
  for (int tap = 0; tap < 9; ++tap) {
    double acc = 0.0;
    for (int c = 0; c < COND_C; ++c) acc += (double)w3[((size_t)co * COND_C + c) * 9 + tap] * (double)e[c];
    etab[((size_t)t * 10 + tap) * HID_C + co] = (float)acc;
    tot += acc;
  }
  etab[((size_t)t * 10 + 9) * HID_C + co] = (float)tot;
}
hipError_t launch_etab(const float* w3_oihw, const float* emb, float* etab, hipStream_t s) {
  hipLaunchKernelGGL(etab_kernel, dim3(EMB_ROWS), dim3(HID_C), 0, s, w3_oihw, emb, etab);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Swin denoiser, hoisted form (dd_kernels.h: SWIN_CONVA_H / SWIN_PRED_H): the time-embedding term pred.0(convB(convA(E[t] on every pixel)))
// without biases.  Every convolution zero-pads its own input, so the term is constant over the image except within three pixels of the
// border, where it depends on the pixel's distances to the borders only: it is evaluated on an R_h x R_w image (R = min(n, 7)), whose
// pixels are the border classes (swin_tt_class), once per plan and parameter generation.  fp64 accumulation, fp32 between the stages.
// ------------------------------------------------------------------------------------------------
// out[k][r][c][co] = sum over the taps inside the R_h x R_w image, sum_ci w[co][ci][tap] * in[k][r+dy][c+dx][ci];  in == NULL: emb[ts[k]][ci]
__global__ void __launch_bounds__(256) swin_tt_conv_kernel(const float* __restrict__ in, const float* __restrict__ emb, const long long* __restrict__ ts,
                                                           const float* __restrict__ wt, float* __restrict__ out, int RH, int RW, int cin, int cout) {
  const int pix = blockIdx.x % (RH * RW), k = blockIdx.x / (RH * RW);
  const int r = pix / RW, c = pix - r * RW, co = threadIdx.x;
  if (co >= cout) return;
  double acc = 0.0;
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = r + ky - 1;
    if (iy < 0 || iy >= RH) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = c + kx - 1;
      if (ix < 0 || ix >= RW) continue;
      const float* ip = in ? in + (((size_t)k * RH + iy) * RW + ix) * cin : emb + (size_t)clamp_t(ts[k]) * COND_C;
      const float* wp = wt + (size_t)co * cin * 9 + ky * 3 + kx;
      for (int ci = 0; ci < cin; ++ci) acc += (double)wp[(size_t)ci * 9] * (double)ip[ci];
    }
  }
  out[((size_t)blockIdx.x) * cout + co] = (float)acc;
}
// ttab[k][0][co] = value of the reference class, ttab[k][1 + 7 r + c][co] = value of class (r, c) minus it (rows of classes the image does not have: 0)
__global__ void __launch_bounds__(64) swin_tt_final_kernel(const float* __restrict__ timg, float* __restrict__ ttab, int RH, int RW) {
  const int k = blockIdx.x, co = threadIdx.x;
  const float* ti = timg + (size_t)k * RH * RW * HID_C;
  float* tt = ttab + (size_t)k * SWIN_TT_ROWS * HID_C;
  const float ref = ti[((size_t)swin_tt_ref(RH) * RW + swin_tt_ref(RW)) * HID_C + co];
  tt[co] = ref;
  for (int r = 0; r < SWIN_TT_AX; ++r)
    for (int c = 0; c < SWIN_TT_AX; ++c)
      tt[(size_t)(1 + r * SWIN_TT_AX + c) * HID_C + co] = (r < RH && c < RW) ? ti[((size_t)r * RW + c) * HID_C + co] - ref : 0.f;
}
hipError_t launch_swin_ttab(const float* wa_oihw, const float* wb_oihw, const float* w3_oihw, const float* emb, const long long* ts, int T,
                            int h, int w, float* scratch, float* ttab, hipStream_t s) {
  const int RH = h < SWIN_TT_AX ? h : SWIN_TT_AX, RW = w < SWIN_TT_AX ? w : SWIN_TT_AX;
  const size_t npix = (size_t)T * RH * RW;
  float *a_img = scratch, *b_img = a_img + npix * COND_C, *t_img = b_img + npix * COND_C;
  hipLaunchKernelGGL(swin_tt_conv_kernel, dim3((unsigned)npix), dim3(COND_C), 0, s, (const float*)nullptr, emb, ts, wa_oihw, a_img, RH, RW, COND_C, COND_C);
  hipLaunchKernelGGL(swin_tt_conv_kernel, dim3((unsigned)npix), dim3(COND_C), 0, s, (const float*)a_img, emb, ts, wb_oihw, b_img, RH, RW, COND_C, COND_C);
  hipLaunchKernelGGL(swin_tt_conv_kernel, dim3((unsigned)npix), dim3(COND_C), 0, s, (const float*)b_img, emb, ts, w3_oihw, t_img, RH, RW, COND_C, HID_C);
  hipLaunchKernelGGL(swin_tt_final_kernel, dim3((unsigned)T), dim3(HID_C), 0, s, (const float*)t_img, ttab, RH, RW);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Swin hoist, 5x5 form (SWIN_PRED5_H, dd_kernels.h): pred.0(convB(a)) at pixel q = sum over taps e, d of W3[e] . WB[d] . a(q + e + d), taps e
// restricted to q + e inside the image (pred.0 zero-pads convB's result).  The 5x5 kernel W5[u] = sum over e + d = u of W3[e] . WB[d] sums ALL
// tap pairs; the difference -- pairs whose e leaves the image, at the pixels on the border -- is the correction of swin_bcorr_kernel.
// ------------------------------------------------------------------------------------------------
// w5[co][ci][u] (u = 5 (ey + dy + 2) + (ex + dx + 2)); one thread per output, fp64 accumulation
__global__ void __launch_bounds__(256) swin_w5_kernel(const float* __restrict__ wb, const float* __restrict__ w3, float* __restrict__ w5) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= HID_C * COND_C * 25) return;
  const int u = i % 25, ci = (i / 25) % COND_C, co = i / (25 * COND_C);
  const int uy = u / 5 - 2, ux = u % 5 - 2;
  double acc = 0.0;
  for (int ey = -1; ey <= 1; ++ey) {
    const int dy = uy - ey;
    if (dy < -1 || dy > 1) continue;
    for (int ex = -1; ex <= 1; ++ex) {
      const int dx = ux - ex;
      if (dx < -1 || dx > 1) continue;
      const float* a = w3 + (size_t)co * COND_C * 9 + (ey + 1) * 3 + (ex + 1);                       // W3[co][cm][e], cm stride 9
      const float* b = wb + (size_t)ci * 9 + (dy + 1) * 3 + (dx + 1);                                 // WB[cm][ci][d], cm stride 256 * 9
      for (int cm = 0; cm < COND_C; ++cm) acc += (double)a[(size_t)cm * 9] * (double)b[(size_t)cm * COND_C * 9];
    }
  }
  w5[i] = (float)acc;
}
// pairp[e][d][ci][co] = sum_cm W3[co][cm][e] * WB[cm][ci][d]; block = (e, d, ci), thread = co
__global__ void __launch_bounds__(64) swin_pair_kernel(const float* __restrict__ wb, const float* __restrict__ w3, float* __restrict__ pairp) {
  const int ci = blockIdx.x % COND_C, ed = blockIdx.x / COND_C, e = ed / 9, d = ed % 9, co = threadIdx.x;
  const float* a = w3 + (size_t)co * COND_C * 9 + e;
  const float* b = wb + (size_t)ci * 9 + d;
  double acc = 0.0;
  for (int cm = 0; cm < COND_C; ++cm) acc += (double)a[(size_t)cm * 9] * (double)b[(size_t)cm * COND_C * 9];
  pairp[((size_t)ed * COND_C + ci) * HID_C + co] = (float)acc;
}
hipError_t launch_swin_compose(const float* wb_oihw, const float* w3_oihw, float* w5_oihw, float* pairp, void* kside, hipStream_t s);
// one block per border pixel (ring index) and image: thread = (cout, quarter of the 256 input channels); the 32 channels of an activation block
// and their 32 weight rows are loaded as one batch (independent loads in flight together), four partial sums per thread, LDS reduction
// The pair sum of ONE border pixel (y, x) of image b by a 256-thread block: thread = (cout, quarter of the 256 input channels); the 32 channels of
// an activation block and their 32 weight rows are loaded as one batch, four partial sums per thread, LDS reduction; the result is returned to the
// threads of wave 0 (thread = cout).  sideways_only: only the taps e that leave the image sideways from a row inside it (the corners' share
// beside the line convolutions below, which cover the taps leaving through the top / bottom row).
template <int EK>
__device__ __forceinline__ float swin_bcorr_pixel(const void* __restrict__ sa, const float* __restrict__ pairp, int b, int y, int x, int h, int w,
                                                  bool sideways_only, float (&red)[4][HID_C]) {
  const int co = threadIdx.x & (HID_C - 1), part = threadIdx.x >> 6;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int e = 0; e < 9; ++e) {
    const int py = y + e / 3 - 1, px = x + e % 3 - 1;
    if (py >= 0 && py < h && px >= 0 && px < w) continue;      // pred.0 tap inside the image: nothing to take out
    if (sideways_only && (py < 0 || py >= h)) continue;
    for (int d = 0; d < 9; ++d) {
      const int sy = py + d / 3 - 1, sx = px + d % 3 - 1;
      if (sy < 0 || sy >= h || sx < 0 || sx >= w) continue;    // convB tap on zero padding
      const float* pp = pairp + ((size_t)(e * 9 + d) * COND_C) * HID_C + co;
#pragma unroll
      for (int k = 0; k < COND_C / ACT_CB / 4; ++k) {
        const int cb = part * (COND_C / ACT_CB / 4) + k;
        const size_t off = act_offset(COND_C, h, w, b, cb * ACT_CB, sy, sx);
        float v[ACT_CB], wv[ACT_CB];
        if constexpr (EK == EK_F32) {
#pragma unroll
          for (int q = 0; q < ACT_CB / 4; ++q) {
            const float4 t = reinterpret_cast<const float4*>(static_cast<const float*>(sa) + off)[q];
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
          }
        } else {
#pragma unroll
          for (int q = 0; q < ACT_CB / 8; ++q) {
            const uint4 t = reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(sa) + off)[q];
            const uint32_t u[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              v[8 * q + 2 * i] = EK == EK_BF16 ? bf16_to_f32(u[i] & 0xFFFFu) : f16_to_f32(u[i] & 0xFFFFu);
              v[8 * q + 2 * i + 1] = EK == EK_BF16 ? bf16_to_f32(u[i] >> 16) : f16_to_f32(u[i] >> 16);
            }
          }
        }
#pragma unroll
        for (int c = 0; c < ACT_CB; ++c) wv[c] = pp[(size_t)(cb * ACT_CB + c) * HID_C];
#pragma unroll
        for (int c = 0; c < ACT_CB; ++c) acc[c & 3] = fmaf(wv[c], v[c], acc[c & 3]);
      }
    }
  }
  red[part][co] = (acc[0] + acc[1]) + (acc[2] + acc[3]);
  __syncthreads();
  return (red[0][co] + red[1][co]) + (red[2][co] + red[3][co]);
}
// every border pixel by pair sums, one block per ring index and image: images with an axis shorter than three pixels only (the line kernel below
// needs three), so the nine 64-KB tap-pair matrices a pixel reads do not matter
template <int EK>
__global__ void __launch_bounds__(256) swin_bcorr_kernel(const void* __restrict__ sa, const float* __restrict__ pairp, float* __restrict__ bcorr, int h, int w) {
  __shared__ float red[4][HID_C];
  const int r = blockIdx.x, b = blockIdx.y;
  int y, x;      // inverse of swin_ring_index
  if (r < w) { y = 0; x = r; }
  else if (r < 2 * w) { y = h - 1; x = r - w; }
  else if (r < 2 * w + (h - 2)) { y = r - 2 * w + 1; x = 0; }
  else { y = r - 2 * w - (h - 2) + 1; x = w - 1; }
  const bool used = y < h && x < w && swin_ring_index(y, x, h, w) == r;      // (an index the image's ring does not use -- h == 1, w == 1 -- stays zero)
  const float v = swin_bcorr_pixel<EK>(sa, pairp, b, used ? y : 0, used ? x : 0, h, w, false, red);
  if (threadIdx.x < HID_C) bcorr[((size_t)b * swin_ring_stride(h, w) + r) * HID_C + threadIdx.x] = used ? v : 0.f;
}

// The same correction as four LINE convolutions on the matrix cores (2-byte kinds: 32x32x16 MFMA; fp32 kind: 32x32x2).  At a pixel of the top row every pred.0 tap of kernel row
// -1 leaves the image, whatever its column, and convB reaches back into row 0 only: the correction is a 1x5 convolution along row 0 with
// K_top[u] = sum over ex + dx = u of P[(-1, ex)][(+1, dx)] (zero padding at the row's ends = convB's own padding); bottom row, left and right
// column (rows 1 .. h-2) likewise.  kside: the four 5-tap kernels in MFMA fragment order [side][u][k-step][cout half][lane] x 8 elements
// (lane (i, g): cout 32 nt + i, channels 16 kk + 8 g ..+7), built by swin_kside_kernel per parameter generation.
// One block = 32 consecutive pixels of one side x 64 couts; its four waves take four k-steps (64 input channels) each -- per tap 4 activation and
// 8 weight fragments in flight, 8 MFMAs -- and wave 0 sums the four partial tiles through LDS.  The last four blocks of the grid are the image's
// corners: the taps that leave sideways (swin_bcorr_pixel), into the four entries behind the ring.
template <int EK>
__global__ void __launch_bounds__(256) swin_bcorr_line_kernel(const void* __restrict__ sa, const uint4* __restrict__ kside, const float* __restrict__ pairp,
                                                              float* __restrict__ bcorr, int h, int w) {
  // channels per MFMA step: 16 in the 2-byte kinds (lane (i, g): 8 g ..+7), 8 in fp32 (mma_step<EK_F32> = four 32x32x2 MFMAs; lane (i, g): 4 g ..+3)
  constexpr int CPS = EK == EK_F32 ? 8 : 16, NSTEP = COND_C / CPS, ESZ = EK == EK_F32 ? 4 : 2;
  __shared__ float red[4][HID_C];
  __shared__ float part_acc[3][64][33];
  const int b = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, g = lane >> 5;
  const int nrow = (w + 31) / 32, ncol = (h - 2 + 31) / 32;
  int side, seg = blockIdx.x;
  if (seg >= 2 * nrow + 2 * ncol) {
    const int cid = seg - (2 * nrow + 2 * ncol);
    const float v = swin_bcorr_pixel<EK>(sa, pairp, b, (cid >> 1) ? h - 1 : 0, (cid & 1) ? w - 1 : 0, h, w, true, red);
    if (threadIdx.x < HID_C) bcorr[((size_t)b * swin_ring_stride(h, w) + swin_ring_size(h, w) + cid) * HID_C + threadIdx.x] = v;
    return;
  }
  if (seg < nrow) side = 0;
  else if (seg < 2 * nrow) { side = 1; seg -= nrow; }
  else if (seg < 2 * nrow + ncol) { side = 2; seg -= 2 * nrow; }
  else { side = 3; seg -= 2 * nrow + ncol; }
  const int L = side < 2 ? w : h;                         // the line: row 0 / h-1, column 0 / w-1
  const int t = (side < 2 ? 0 : 1) + seg * 32 + li;       // this lane's pixel along it (columns: rows 1 .. h-2 only, the corners belong to the rows)
  f32x16_t acc[2];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[n][i] = 0.f;
  const uint4* ks = kside + (size_t)side * 5 * NSTEP * 2 * 64 + lane;
#pragma unroll 1
  for (int u = 0; u < 5; ++u) {
    const int tp = t + u - 2;
    const bool valid = tp >= 0 && tp < L;
    const int tc = valid ? tp : 0;
    const int sy = side == 0 ? 0 : side == 1 ? h - 1 : tc, sx = side == 2 ? 0 : side == 3 ? w - 1 : tc;
#pragma unroll
    for (int k4 = 0; k4 < NSTEP / 4; ++k4) {
      const int kk = wave * (NSTEP / 4) + k4;
      uint4 pf = *reinterpret_cast<const uint4*>(static_cast<const char*>(sa) + act_offset(COND_C, h, w, b, kk * CPS + g * (CPS / 2), sy, sx) * ESZ);
      if (!valid) pf = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
      for (int n = 0; n < 2; ++n) mma_step<EK>(acc[n], ks[((size_t)(u * NSTEP + kk) * 2 + n) * 64], pf);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int i = 0; i < 16; ++i) part_acc[wave - 1][lane][n * 16 + i] = acc[n][i];
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[n][i] += (part_acc[0][lane][n * 16 + i] + part_acc[1][lane][n * 16 + i]) + part_acc[2][lane][n * 16 + i];
  const bool store = side < 2 ? t < w : t <= h - 2;
  if (store) {
    const int ridx = side == 0 ? t : side == 1 ? w + t : side == 2 ? 2 * w + (t - 1) : 2 * w + (h - 2) + (t - 1);
    float* dst = bcorr + ((size_t)b * swin_ring_stride(h, w) + ridx) * HID_C;
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<float4*>(dst + n * 32 + 8 * q + 4 * g) = make_float4(acc[n][q * 4], acc[n][q * 4 + 1], acc[n][q * 4 + 2], acc[n][q * 4 + 3]);
  }
}
// kside[kind][side][u][kk][nt][lane][j] (16-bit elements; kind 0 = bf16, 1 = f16) from the tap-pair products: one thread per element
// ... then the fp32 section [side][u][k-step of 8 channels][nt][lane] x 4 floats (lane (i, g): cout 32 nt + i, channels 8 kk + 4 g ..+3)
__global__ void __launch_bounds__(256) swin_kside_kernel(const float* __restrict__ pairp, uint16_t* __restrict__ kside) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  constexpr int PER_KIND = 4 * 5 * 16 * 2 * 64 * 8;      // elements of a 16-bit section = 4 sides x 5 taps x 256 channels x 64 couts
  if (i >= 3 * PER_KIND) return;
  int kind, side, u, co, ci;
  if (i < 2 * PER_KIND) {
    const int j = i & 7, lane = (i >> 3) & 63, nt = (i >> 9) & 1, kk = (i >> 10) & 15, us = (i >> 14) % 20;
    kind = i / PER_KIND; u = us % 5; side = us / 5;
    co = nt * 32 + (lane & 31); ci = kk * 16 + (lane >> 5) * 8 + j;
  } else {
    const int f = i - 2 * PER_KIND;
    const int j = f & 3, lane = (f >> 2) & 63, nt = (f >> 8) & 1, kk = (f >> 9) & 31, us = f >> 14;
    kind = 2; u = us % 5; side = us / 5;
    co = nt * 32 + (lane & 31); ci = kk * 8 + (lane >> 5) * 4 + j;
  }
  float v = 0.f;
  for (int a = -1; a <= 1; ++a) {        // a: the pred.0 tap's free coordinate, c = (u - 2) - a: convB's
    const int c = (u - 2) - a;
    if (c < -1 || c > 1) continue;
    int e, d;
    if (side == 0) { e = 0 * 3 + (a + 1); d = 2 * 3 + (c + 1); }            // e = (-1, a), d = (+1, c)
    else if (side == 1) { e = 2 * 3 + (a + 1); d = 0 * 3 + (c + 1); }       // e = (+1, a), d = (-1, c)
    else if (side == 2) { e = (a + 1) * 3 + 0; d = (c + 1) * 3 + 2; }       // e = (a, -1), d = (c, +1)
    else { e = (a + 1) * 3 + 2; d = (c + 1) * 3 + 0; }                      // e = (a, +1), d = (c, -1)
    v += pairp[((size_t)(e * 9 + d) * COND_C + ci) * HID_C + co];
  }
  if (kind == 2) reinterpret_cast<float*>(kside + 2 * PER_KIND)[i - 2 * PER_KIND] = v;
  else kside[i] = (uint16_t)(kind == 0 ? f32_to_bf16(v) : f32_to_f16(v));
}
hipError_t launch_swin_compose(const float* wb_oihw, const float* w3_oihw, float* w5_oihw, float* pairp, void* kside, hipStream_t s) {
  hipLaunchKernelGGL(swin_w5_kernel, dim3((HID_C * COND_C * 25 + 255) / 256), dim3(256), 0, s, wb_oihw, w3_oihw, w5_oihw);
  hipLaunchKernelGGL(swin_pair_kernel, dim3(81 * COND_C), dim3(HID_C), 0, s, wb_oihw, w3_oihw, pairp);
  hipLaunchKernelGGL(swin_kside_kernel, dim3((unsigned)(3 * (SWIN_KSIDE_BYTES / 8) + 255) / 256), dim3(256), 0, s, (const float*)pairp, static_cast<uint16_t*>(kside));
  return hipGetLastError();
}
hipError_t launch_swin_bcorr(const void* sa, int ek, const float* pairp, const void* kside, float* bcorr, int B, int h, int w, hipStream_t s) {
  if ((ek == EK_BF16 || ek == EK_F16 || ek == EK_F32) && h >= 3 && w >= 3) {
    // matrix-core line convolutions + the corners' sideways taps, one launch; kside sections: bf16, f16 (a quarter of the bytes each), fp32 (half)
    const dim3 grid((unsigned)(2 * ((w + 31) / 32) + 2 * ((h - 2 + 31) / 32) + 4), (unsigned)B);
    const uint4* ks = static_cast<const uint4*>(kside) + (ek == EK_BF16 ? 0 : ek == EK_F16 ? SWIN_KSIDE_BYTES / 64 : SWIN_KSIDE_BYTES / 32);
    if (ek == EK_BF16) hipLaunchKernelGGL(swin_bcorr_line_kernel<EK_BF16>, grid, dim3(256), 0, s, sa, ks, pairp, bcorr, h, w);
    else if (ek == EK_F16) hipLaunchKernelGGL(swin_bcorr_line_kernel<EK_F16>, grid, dim3(256), 0, s, sa, ks, pairp, bcorr, h, w);
    else hipLaunchKernelGGL(swin_bcorr_line_kernel<EK_F32>, grid, dim3(256), 0, s, sa, ks, pairp, bcorr, h, w);
    return hipGetLastError();
  }
  // (the four entries behind the ring stay zero: cleared when the plan allocated the buffer)
  const dim3 grid((unsigned)swin_ring_size(h, w), (unsigned)B);
  if (ek == EK_F32) hipLaunchKernelGGL(swin_bcorr_kernel<EK_F32>, grid, dim3(256), 0, s, sa, pairp, bcorr, h, w);
  else if (ek == EK_BF16) hipLaunchKernelGGL(swin_bcorr_kernel<EK_BF16>, grid, dim3(256), 0, s, sa, pairp, bcorr, h, w);
  else if (ek == EK_F16) hipLaunchKernelGGL(swin_bcorr_kernel<EK_F16>, grid, dim3(256), 0, s, sa, pairp, bcorr, h, w);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// q_sample (reference scheduling_ddim.py:355-376): out = sqrt(abar_t)*x0 + sqrt(1-abar_t)*noise
// ------------------------------------------------------------------------------------------------
__global__ void add_noise_kernel(const float* __restrict__ x0, const float* __restrict__ noise,
                                 const long long* __restrict__ t, const float* __restrict__ acp, int n_train,
                                 float* __restrict__ out, long long per_sample) {
  const int b = blockIdx.y;
  long long tb = t[b];
  tb = tb < 0 ? 0 : (tb >= n_train ? n_train - 1 : tb);
  const float a = acp[tb];
  const float sa = sqrtf(a), sb = sqrtf(1.f - a);     // fp32 pow(.,0.5) of the fp32 table, as the reference
  const long long i4 = ((long long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= per_sample) return;
  const size_t o = (size_t)b * per_sample + i4;
  if (i4 + 4 <= per_sample && (per_sample & 3) == 0) {
    const float4 xv = *reinterpret_cast<const float4*>(x0 + o), nv = *reinterpret_cast<const float4*>(noise + o);
    *reinterpret_cast<float4*>(out + o) = make_float4(sa * xv.x + sb * nv.x, sa * xv.y + sb * nv.y,
                                                      sa * xv.z + sb * nv.z, sa * xv.w + sb * nv.w);
  } else {
    for (long long k = i4; k < per_sample && k < i4 + 4; ++k) {
      const size_t oo = (size_t)b * per_sample + k;
      out[oo] = sa * x0[oo] + sb * noise[oo];
    }
  }
}
hipError_t launch_add_noise(const float* x0, const float* noise, const long long* t, const float* acp, int n_train,
                            float* out, int B, long long per_sample, hipStream_t s) {
  dim3 grid((unsigned)((per_sample / 4 + 256) / 256), (unsigned)B);
  hipLaunchKernelGGL(add_noise_kernel, grid, dim3(256), 0, s, x0, noise, t, acp, n_train, out, per_sample);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// latent encoder  (reference src/model/ops/depth_transform.py:15-19, eval-mode BN folded on the host)
//   stage 0: conv 1->16, 3x3, stride 2, pad 1 (+BN) + LeakyReLU(0.2)     -> tmp NHWC [B][h][w][16]
//   stage 1: conv 16->16, 3x3, pad 1 (+BN) + tanh                         -> latent NCHW
// one thread per latent pixel, all 16 channels in registers; weights are wave-uniform (scalar loads)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) enc0_kernel(const float* __restrict__ depth, const float* __restrict__ w0,
                                                   const float* __restrict__ b0, float* __restrict__ tmp,
                                                   int H, int W, int h, int w) {
  const int b = blockIdx.y;
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (long long)h * w) return;
  const int oy = (int)(p / w), ox = (int)(p - (long long)oy * w);
  const float* d = depth + (size_t)b * H * W;
  float in[9];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
      in[ky * 3 + kx] = (iy >= 0 && iy < H && ix >= 0 && ix < W) ? d[(size_t)iy * W + ix] : 0.f;
    }
  float o[LATENT_C];
#pragma unroll
  for (int c = 0; c < LATENT_C; ++c) {
    float a = b0[c];
#pragma unroll
    for (int k = 0; k < 9; ++k) a = fmaf(w0[c * 9 + k], in[k], a);
    o[c] = a >= 0.f ? a : 0.2f * a;
  }
  float4* dst = reinterpret_cast<float4*>(tmp + ((size_t)b * h * w + p) * LATENT_C);
#pragma unroll
  for (int q = 0; q < 4; ++q) dst[q] = make_float4(o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]);
}

// Stage 1, round 4: weights TAP-MAJOR ([tap][ci][co]: CodecWeights::enc_w1t) so that a (tap, ci) row of 16 couts is ONE 64-byte scalar load (the OIHW
// table cost one 4-byte scalar load per FMA: 257 s_load_dword per tap); cout PAIRS on v_pk_fma_f32 (two fp32 FMAs per lane and issue slot; the plain
// v_fma_f32 runs at half the vector unit's rate); the NEXT tap's 64 input bytes are requested before the current tap's FMAs.  A tap outside the image
// contributes exact zeros (clamped address, value selected to 0) instead of being skipped; per output the order of the accumulation is unchanged (tap,
// then ci) and a packed FMA is two IEEE FMAs: the same bits as before.
__global__ void __launch_bounds__(256) enc1_kernel(const float* __restrict__ tmp, const float* __restrict__ w1t,
                                                   const float* __restrict__ b1, float* __restrict__ latent, int h, int w) {
  const int b = blockIdx.y;
  const long long HW = (long long)h * w;
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool ok = p < HW;
  const long long pc = ok ? p : HW - 1;
  const int oy = (int)(pc / w), ox = (int)(pc - (long long)oy * w);
  f32x2_t acc[LATENT_C / 2];
#pragma unroll
  for (int c = 0; c < LATENT_C / 2; ++c) acc[c] = (f32x2_t){b1[2 * c], b1[2 * c + 1]};
  const float* img = tmp + (size_t)b * HW * LATENT_C;
  float4 cur[4], nx[4];
  bool cur_in = false, nx_in = false;
  auto fetch = [&](int tap, float4 (&dst)[4], bool& inside) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const int iy = oy - 1 + ky, ix = ox - 1 + kx;
    inside = iy >= 0 && iy < h && ix >= 0 && ix < w;
    const int iyc = iy < 0 ? 0 : (iy >= h ? h - 1 : iy), ixc = ix < 0 ? 0 : (ix >= w ? w - 1 : ix);
    const float4* src = reinterpret_cast<const float4*>(img + ((size_t)iyc * w + ixc) * LATENT_C);
#pragma unroll
    for (int q = 0; q < 4; ++q) dst[q] = src[q];
  };
  fetch(0, cur, cur_in);
#pragma unroll 1
  for (int tap = 0; tap < 9; ++tap) {
    fetch(tap < 8 ? tap + 1 : 8, nx, nx_in);            // (behind the last tap: tap 8 once more, discarded)
    float in[LATENT_C];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      in[4 * q] = cur_in ? cur[q].x : 0.f; in[4 * q + 1] = cur_in ? cur[q].y : 0.f; in[4 * q + 2] = cur_in ? cur[q].z : 0.f; in[4 * q + 3] = cur_in ? cur[q].w : 0.f;
    }
    const float* wt = w1t + (size_t)tap * LATENT_C * LATENT_C;      // wave-uniform
    // channel groups of four, every index a compile-time constant; the scheduling fence keeps a group's four scalar rows from being hoisted above
    // the previous group's FMAs (16 rows in flight = 256 SGPRs)
#pragma unroll
    for (int q = 0; q < LATENT_C / 4; ++q) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int ci = 4 * q + j;
        const f32x2_t* row = reinterpret_cast<const f32x2_t*>(wt + ci * LATENT_C);
        const f32x2_t a = (f32x2_t){in[ci], in[ci]};
#pragma unroll
        for (int co = 0; co < LATENT_C / 2; ++co) acc[co] = __builtin_elementwise_fma(row[co], a, acc[co]);
      }
      DD_SCHED_FENCE();
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) cur[q] = nx[q];
    cur_in = nx_in;
  }
  if (ok) {
#pragma unroll
    for (int c = 0; c < LATENT_C; ++c) latent[((size_t)b * LATENT_C + c) * HW + p] = tanhf(acc[c >> 1][c & 1]);
  }
}

hipError_t launch_encode(const CodecWeights& cw, const float* depth, float* tmp_nhwc, float* latent_nchw,
                         int B, int H, int W, hipStream_t s) {
  const int h = (H - 1) / 2 + 1, w = (W - 1) / 2 + 1;
  dim3 grid((unsigned)(((long long)h * w + 255) / 256), (unsigned)B);
  hipLaunchKernelGGL(enc0_kernel, grid, dim3(256), 0, s, depth, cw.enc_w0, cw.enc_b0, tmp_nhwc, H, W, h, w);
  hipLaunchKernelGGL(enc1_kernel, grid, dim3(256), 0, s, tmp_nhwc, cw.enc_w1t, cw.enc_b1, latent_nchw, h, w);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// latent decoder  (reference src/model/ops/depth_transform.py:20-26,33-35, eval-mode BN folded)
//   stage 0: ConvTranspose2d 16->16, k4 s2 p1 (+bias, BN) + ReLU -> tmp NHWC [B][2h][2w][16]
//            out[oy][ox] = sum over ky with (oy+1-ky) even, iy=(oy+1-ky)/2 in range (same in x): 2x2 taps
//   stage 1: conv 16->1 3x3 pad 1 (+bias) -> sigmoid -> 1/max(s,1e-6) - 1   (fp32 throughout)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) dec0_kernel(const float* __restrict__ latent, const float* __restrict__ w0,
                                                   const float* __restrict__ b0, float* __restrict__ tmp, int h, int w) {
  const int b = blockIdx.y;
  const int H = 2 * h, W = 2 * w;
  const long long HW = (long long)h * w;
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (long long)H * W) return;
  const int oy = (int)(p / W), ox = (int)(p - (long long)oy * W);
  float acc[LATENT_C];
#pragma unroll
  for (int c = 0; c < LATENT_C; ++c) acc[c] = b0[c];
  const float* lat = latent + (size_t)b * LATENT_C * HW;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int ky = ((oy + 1) & 1) + 2 * a;
    const int iy = (oy + 1 - ky) >> 1;
    if (iy < 0 || iy >= h) continue;
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
      const int kx = ((ox + 1) & 1) + 2 * c2;
      const int ix = (ox + 1 - kx) >> 1;
      if (ix < 0 || ix >= w) continue;
      for (int ci = 0; ci < LATENT_C; ++ci) {
        const float v = lat[(size_t)ci * HW + (size_t)iy * w + ix];
        const float* wr = w0 + ((size_t)ci * LATENT_C) * 16 + ky * 4 + kx;
#pragma unroll
        for (int co = 0; co < LATENT_C; ++co) acc[co] = fmaf(wr[co * 16], v, acc[co]);
      }
    }
  }
  float4* dst = reinterpret_cast<float4*>(tmp + ((size_t)b * H * W + p) * LATENT_C);
#pragma unroll
  for (int q = 0; q < 4; ++q)
    dst[q] = make_float4(relu_keep_nan(acc[4 * q]), relu_keep_nan(acc[4 * q + 1]), relu_keep_nan(acc[4 * q + 2]), relu_keep_nan(acc[4 * q + 3]));
}

__global__ void __launch_bounds__(256) dec1_kernel(const float* __restrict__ tmp, const float* __restrict__ w1, float b1,
                                                   float* __restrict__ depth, int H, int W) {
  const int b = blockIdx.y;
  const long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= (long long)H * W) return;
  const int oy = (int)(p / W), ox = (int)(p - (long long)oy * W);
  float z = b1;
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy - 1 + ky;
    if (iy < 0 || iy >= H) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox - 1 + kx;
      if (ix < 0 || ix >= W) continue;
      const float4* src = reinterpret_cast<const float4*>(tmp + ((size_t)b * H * W + (size_t)iy * W + ix) * LATENT_C);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = src[q];
        const float* wk = w1 + (4 * q) * 9 + ky * 3 + kx;
        z = fmaf(wk[0], v.x, z); z = fmaf(wk[9], v.y, z); z = fmaf(wk[18], v.z, z); z = fmaf(wk[27], v.w, z);
      }
    }
  }
  const float sg = 1.f / (1.f + expf(-z));
  depth[(size_t)b * H * W + p] = 1.f / ((sg < 1e-6f) ? 1e-6f : sg) - 1.f;
}

// Fused decoder: one workgroup produces a 14x30 tile of the depth map.  The (7+2)x(15+2) latent patch and the ReLU'd
// transposed-convolution output of the 16x32 region (the 3x3 conv's halo) live in LDS, so the 16-channel full
// resolution intermediate (110 MB per B=4 KITTI batch) never goes to HBM.  Wave w computes the region pixels of output
// parity class w = ((oy+1)&1, (ox+1)&1): its 2x2 live taps of the 4x4 kernel are wave-uniform, so the weights are scalar
// operands (tap-major copy dec_w0t).
// Round 4: the tile was 16x32 (region 18x34: 153 pixels per class = three passes of a 64-lane wave, the last one 39 % full) and a lane computed ONE
// pixel per pass -- every 16 FMAs waited for a 64-byte scalar weight load (s_load results return out of order: the only wait is lgkmcnt(0), so the
// load cannot be pipelined across iterations).  Now the region is 16x32 = 128 pixels per class = ONE pass with TWO pixels per lane: 32 FMAs per
// weight row, a third of the loop iterations.  Same accumulation order per output: bit-identical results.
constexpr int DT_H = 14, DT_W = 30;                 // output tile
constexpr int DR_H = DT_H + 2, DR_W = DT_W + 2;     // region of the intermediate (halo 1)
constexpr int DL_H = DT_H / 2 + 2, DL_W = DT_W / 2 + 2, DL_P = DL_W + 1;   // latent patch rows / cols / padded row pitch
__global__ void __launch_bounds__(256) dec_fused_kernel(const float* __restrict__ latent, const float* __restrict__ w0t,
                                                        const float* __restrict__ b0, const float* __restrict__ w1, float b1,
                                                        float* __restrict__ depth, int h, int w) {
  __shared__ float s_lat[LATENT_C][DL_H][DL_P];
  __shared__ __attribute__((aligned(16))) float s_mid[DR_H][DR_W][LATENT_C];
  __shared__ __attribute__((aligned(16))) float s_w1[9][LATENT_C];     // 3x3 conv weights, tap-major (144 scalars would spill the SGPR file)
  const int b = blockIdx.z;
  const int H = 2 * h, W = 2 * w;
  const int Y0 = blockIdx.y * DT_H, X0 = blockIdx.x * DT_W;
  const int iy0 = Y0 / 2 - 1, ix0 = X0 / 2 - 1;                 // latent coordinates of patch (0, 0)
  const int tid = threadIdx.x;
  const long long HWl = (long long)h * w;
  const float* lat = latent + (size_t)b * LATENT_C * HWl;
  if (tid < 9 * LATENT_C) s_w1[tid / LATENT_C][tid % LATENT_C] = w1[(tid % LATENT_C) * 9 + tid / LATENT_C];
  for (int i = tid; i < LATENT_C * DL_H * DL_W; i += 256) {
    const int c = i / (DL_H * DL_W), rem = i - c * (DL_H * DL_W);
    const int r = rem / DL_W, q = rem - r * DL_W;
    const int iy = iy0 + r, ix = ix0 + q;
    s_lat[c][r][q] = (iy >= 0 && iy < h && ix >= 0 && ix < w) ? lat[(size_t)c * HWl + (size_t)iy * w + ix] : 0.f;
  }
  __syncthreads();
  // ---- transposed conv (k4 s2 p1) + bias + ReLU over the region; region (r, c) <-> output (Y0 - 1 + r, X0 - 1 + c).
  //      Y0, X0 are even, so (oy + 1) & 1 == r & 1: live taps ky = (r & 1) + {0, 2}, latent patch row (r - ky) / 2 + 1.
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  const int pr = wave >> 1, pc = wave & 1;                      // this wave's parity class
  constexpr int NR = DR_H / 2, NC = DR_W / 2;                   // 9 x 17 pixels per class
  static_assert((NR * NC) % 128 == 0 && 64 % NC == 0, "a pass = two pixels per lane, the second one 64 / NC class rows below the first");
  constexpr int RSTEP = 64 / NC;                                // class rows between a lane's two pixels
  for (int idx = lane; idx < NR * NC; idx += 128) {
    const int rr = idx / NC, cc = idx - rr * NC;
    const int r = 2 * rr + pr, c = 2 * cc + pc;                 // second pixel: (r + 2 RSTEP, c)
    // accumulators as cout PAIRS: v_pk_fma_f32 does two fp32 FMAs per lane and issue slot (the plain v_fma_f32 runs at half the vector unit's rate);
    // weights = an SGPR pair of the scalar row, the pixel value broadcast to both halves (op_sel).  Two IEEE FMAs: the same bits as fmaf.
    f32x2_t acc[2][LATENT_C / 2];
#pragma unroll
    for (int co = 0; co < LATENT_C / 2; ++co) acc[0][co] = acc[1][co] = (f32x2_t){b0[2 * co], b0[2 * co + 1]};
    // (tap and channel loops stay rolled: each (tap, ci) row of 16 weights is one s_load_dwordx16; unrolling them made the
    //  compiler hold 256 scalar weights and spill SGPRs through v_readlane / v_writelane)
#pragma unroll 1
    for (int a = 0; a < 2; ++a) {
      const int ky = pr + 2 * a;
      const int lr = (r - ky) / 2 + 1;                          // exact: r - ky is even
#pragma unroll 1
      for (int d = 0; d < 2; ++d) {
        const int kx = pc + 2 * d;
        const int lc = (c - kx) / 2 + 1;
        const float* wt = w0t + (size_t)(ky * 4 + kx) * LATENT_C * LATENT_C;     // wave-uniform
#pragma unroll 4
        for (int ci = 0; ci < LATENT_C; ++ci) {
          const float v0 = s_lat[ci][lr][lc], v1 = s_lat[ci][lr + RSTEP][lc];
          const f32x2_t* row = reinterpret_cast<const f32x2_t*>(wt + ci * LATENT_C);
#pragma unroll
          for (int co = 0; co < LATENT_C / 2; ++co) {
            const f32x2_t wv = row[co];
            acc[0][co] = __builtin_elementwise_fma(wv, (f32x2_t){v0, v0}, acc[0][co]);
            acc[1][co] = __builtin_elementwise_fma(wv, (f32x2_t){v1, v1}, acc[1][co]);
          }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int rk = r + 2 * RSTEP * k;
      const int oy = Y0 - 1 + rk, ox = X0 - 1 + c;
      const bool inside = oy >= 0 && oy < H && ox >= 0 && ox < W;   // the 3x3 conv zero-pads OUTSIDE the image
      float4* dst = reinterpret_cast<float4*>(&s_mid[rk][c][0]);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        dst[q] = inside ? make_float4(relu_keep_nan(acc[k][2 * q][0]), relu_keep_nan(acc[k][2 * q][1]), relu_keep_nan(acc[k][2 * q + 1][0]), relu_keep_nan(acc[k][2 * q + 1][1]))
                        : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  __syncthreads();
  // ---- conv 16 -> 1 3x3 (+bias) -> sigmoid -> 1 / max(s, 1e-6) - 1 ----
  for (int i = tid; i < DT_H * DT_W; i += 256) {
    const int ty = i / DT_W, tx = i - ty * DT_W;
    const int oy = Y0 + ty, ox = X0 + tx;
    if (oy >= H || ox >= W) continue;
    float z = b1;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const float4* src = reinterpret_cast<const float4*>(&s_mid[ty + ky][tx + kx][0]);
        const float4* wk = reinterpret_cast<const float4*>(&s_w1[ky * 3 + kx][0]);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = src[q], wv = wk[q];
          z = fmaf(wv.x, v.x, z); z = fmaf(wv.y, v.y, z); z = fmaf(wv.z, v.z, z); z = fmaf(wv.w, v.w, z);
        }
      }
    const float sg = 1.f / (1.f + expf(-z));
    depth[(size_t)b * H * W + (size_t)oy * W + ox] = 1.f / ((sg < 1e-6f) ? 1e-6f : sg) - 1.f;
  }
}

hipError_t launch_decode(const CodecWeights& cw, const float* latent_nchw, float* tmp_nhwc, float* depth,
                         int B, int h, int w, hipStream_t s) {
  const int H = 2 * h, W = 2 * w;
  if (cw.dec_w0t != nullptr) {
    (void)tmp_nhwc;
    dim3 grid((unsigned)((W + DT_W - 1) / DT_W), (unsigned)((H + DT_H - 1) / DT_H), (unsigned)B);
    hipLaunchKernelGGL(dec_fused_kernel, grid, dim3(256), 0, s, latent_nchw, cw.dec_w0t, cw.dec_b0, cw.dec_w1, cw.dec_b1, depth, h, w);
    return hipGetLastError();
  }
  dim3 grid((unsigned)(((long long)H * W + 255) / 256), (unsigned)B);
  hipLaunchKernelGGL(dec0_kernel, grid, dim3(256), 0, s, latent_nchw, cw.dec_w0, cw.dec_b0, tmp_nhwc, h, w);
  hipLaunchKernelGGL(dec1_kernel, grid, dim3(256), 0, s, tmp_nhwc, cw.dec_w1, cw.dec_b1, depth, H, W);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Weights into kernel layout ON THE DEVICE (training: the optimizer updates the fp32 parameters in HBM every step; repacking them
// on the host costs a D2H copy per tensor plus ~25 ns per packed element, single-threaded, for every element kind and kernel
// family -- hundreds of ms per iteration).  One thread per PACKED element; same index map, zero padding, LDS swizzle and
// round-to-nearest-even conversions as the host packer pack_conv_weights() in dd_api.cpp (the two are compared bit for bit in
// tests/test_library_host_emulation.py).
//   transposed == 0: src = W[cout][cin][ks][ks]
//   transposed == 1: the data-gradient convolution of W: value(co, ci, dy, dx) = W[ci][co][ks-1-dy][ks-1-dx], W = [g.cin][g.cout][ks][ks]
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pack_weights_kernel(const float* __restrict__ src, void* __restrict__ dst, PackGeom g, int ek,
                                                            int swizzle, int transposed, long long n_el) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_el) return;
  const int ks = g.ks, kk = ks * ks, n_chunks = g.cin / g.ck, n_tg = kk / g.tg;
  const int esz = ek == EK_F32 ? 4 : 2;
  const int rowb = g.ck * esz, ppp = rowb / 16, rpb = 256 / rowb, epp = 16 / esz;
  // idx = ((((nt * n_chunks + ch) * n_tg + tg) * g.tg) * g.nt * g.ck)  +  row * g.ck + piece_sw * epp + within,  row = t * g.nt + n
  // split f16 (g.planes == 2): a stage block is [hi plane | lo plane], each laid out as above
  const long long plane = (long long)g.tg * g.nt * g.ck;
  const long long blk = plane * (g.planes > 1 ? 2 : 1);
  const long long outer = idx / blk;
  int inner = (int)(idx - outer * blk);
  const bool lo_plane = inner >= plane;
  if (lo_plane) inner -= (int)plane;
  const int tg = (int)(outer % n_tg);
  const int ch = (int)((outer / n_tg) % n_chunks);
  const int nt = (int)(outer / ((long long)n_tg * n_chunks));
  const int row = inner / g.ck, col = inner - row * g.ck;
  const int piece_sw = col / epp, within = col - piece_sw * epp;
  const int piece = swizzle ? (piece_sw ^ ((row / rpb) & (ppp - 1))) : piece_sw;      // the XOR is its own inverse
  const int k = piece * epp + within;
  const int t = row / g.nt, n = row - t * g.nt;
  const int tap = tg * g.tg + t, co = nt * g.nt + n, ci = ch * g.ck + k;
  float v = 0.f;
  if (co < g.cout)
    v = transposed ? src[((size_t)ci * g.cout + co) * kk + (kk - 1 - tap)] : src[((size_t)co * g.cin + ci) * kk + tap];
  if (g.stack && co >= g.cout && co < 2 * g.cout) {
    // stacked image (conv4, EK_F16R): row cout + c = the lo half of row c, times STACK_LSCALE -- the host packer's arithmetic
    const float wv = src[((size_t)(co - g.cout) * g.cin + ci) * kk + tap];
    reinterpret_cast<uint16_t*>(dst)[idx] = (uint16_t)f32_to_f16((wv - f16_to_f32(f32_to_f16(wv))) * STACK_LSCALE);
    return;
  }
  if (g.planes > 1) {
    // hi = f16(w * SPLIT_WSCALE), lo = f16(w * SPLIT_WSCALE - hi): the arithmetic of the host packer (pack_conv_weights in dd_api.cpp)
    const float vs = v * SPLIT_WSCALE;
    const uint32_t hi = f32_to_f16(vs);
    reinterpret_cast<uint16_t*>(dst)[idx] = (uint16_t)(lo_plane ? f32_to_f16(vs - f16_to_f32(hi)) : hi);
  } else if (ek == EK_F32) reinterpret_cast<float*>(dst)[idx] = v;
  else reinterpret_cast<uint16_t*>(dst)[idx] = (uint16_t)(ek == EK_BF16 ? f32_to_bf16(v) : f32_to_f16(v));
}

size_t pack_weights_bytes(const PackGeom& g, int ek) {
  return (size_t)(g.cout_pad / g.nt) * (g.cin / g.ck) * (size_t)(g.ks * g.ks) * g.nt * g.ck * (ek == EK_F32 ? 4 : 2) * (g.planes > 1 ? 2 : 1);
}

hipError_t launch_pack_weights(const float* src, void* dst, const PackGeom& g, int ek, bool swizzle, bool transposed, hipStream_t s) {
  const long long n_el = (long long)(pack_weights_bytes(g, ek) / (ek == EK_F32 ? 4 : 2));
  hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)((n_el + 255) / 256)), dim3(256), 0, s, src, dst, g, ek, swizzle ? 1 : 0,
                     transposed ? 1 : 0, n_el);
  return hipGetLastError();
}

// max |w| of a tensor, as the bits of a non-negative float (atomicMax on unsigned keeps the order): the device route's twin of the host
// packer's "fits the split-f16 image" check (dd_api.cpp: |w| x SPLIT_WSCALE must stay below f16's range)
__global__ void __launch_bounds__(256) max_abs_kernel(const float* __restrict__ w, long long n, unsigned* __restrict__ out) {
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const float a = fabsf(w[i]);
    m = (a > m || a != a) ? a : m;            // a NaN weight counts as "does not fit" (its bit pattern is above every finite value's)
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { const float o = __shfl_xor(m, off, 64); m = (o > m || o != o) ? o : m; }
  if ((threadIdx.x & 63) == 0) atomicMax(out, __builtin_bit_cast(unsigned, m));
}
// debug (option "check_finite"): number of non-finite elements of a tensor (kind 0 fp32, 1 bf16, 2 f16, 3 fp64) added to *out
__global__ void __launch_bounds__(256) count_nonfinite_kernel(const void* __restrict__ p, long long n, int kind, unsigned* __restrict__ out) {
  unsigned c = 0;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    float v;
    if (kind == 0) v = static_cast<const float*>(p)[i];
    else if (kind == 1) v = bf16_to_f32(static_cast<const uint16_t*>(p)[i]);
    else if (kind == 2) v = f16_to_f32(static_cast<const uint16_t*>(p)[i]);
    else { const double d = static_cast<const double*>(p)[i]; v = (d - d == 0.0) ? 0.f : __builtin_nanf(""); }
    if (!(v - v == 0.f)) ++c;
  }
  if (c) atomicAdd(out, c);
}
// One wavefront that does nothing for `ticks` of the constant 100-MHz clock: the two halves of the lane-overlap probe (dd_api.cpp: acquire_lane_stream)
#ifndef DD_HOST_EMULATION
__global__ void spin_kernel(long long ticks, long long* __restrict__ stamp) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
  if (stamp && threadIdx.x == 0) { stamp[0] = t0; stamp[1] = wall_clock64(); }
}
hipError_t launch_spin(long long ticks, long long* stamp, hipStream_t s) {
  hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, ticks, stamp);
  return hipGetLastError();
}
#else
hipError_t launch_spin(long long, long long*, hipStream_t) { return hipSuccess; }
#endif

hipError_t launch_count_nonfinite(const void* p, long long n, int kind, unsigned* out, hipStream_t s) {
  const unsigned nb = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(count_nonfinite_kernel, dim3(nb < 1 ? 1 : (nb > 4096 ? 4096 : nb)), dim3(256), 0, s, p, n, kind, out);
  return hipGetLastError();
}
hipError_t launch_max_abs(const float* w, long long n, unsigned* out_bits, hipStream_t s) {
  const unsigned nb = (unsigned)((n + 256 * 8 - 1) / (256 * 8));
  hipLaunchKernelGGL(max_abs_kernel, dim3(nb < 1 ? 1 : (nb > 1024 ? 1024 : nb)), dim3(256), 0, s, w, n, out_bits);
  return hipGetLastError();
}

// EK_F16R: conv3(cond) from the accumulator-fragment order of 8x32 tiles (fp32; what the split-f16 layer 8 writes: entry
// ((((tile * 4 + wave) * 2 + n) * 2 + m) * 4 + q) * 64 + lane = couts 32n + 8q + 4g .. +3 of pixel (8 ty + 2 wave + m, 32 tx + li), lane = 32g + li)
// into the order of the loop's conv3 tiles -- 8x32 (WM = 2) or 16x32 (WM = 4: pixel row 16 ty + 4 wave + m) -- as f16 quads (out_kind 1) or int16 quads
// with one fp32 scale per (tile, wave, n, m) block of 32 pixels x 32 couts (out_kind 2: max |.| of the block / 32767).  One workgroup of 256 threads
// per destination block (thread = (q, lane)); rows of a 16-row tile below the source's last tile row read as zero (never used: outside the image).
__global__ void __launch_bounds__(256) cadd_reformat_kernel(const float4* __restrict__ src, void* __restrict__ dst, float* __restrict__ scales, int B, int h, int w,
                                                            int big, int out_kind) {
  __shared__ float s_max[4];
  const int WM = big ? 4 : 2, TH = big ? 16 : 8;
  const int tiles_x = (w + 31) / 32, tiles_y = (h + TH - 1) / TH, stiles_y = (h + 7) / 8;
  const long long blk = blockIdx.x;                   // ((tile * 4 + wave) * 2 + n) * WM + m
  const long long e = blk * 256 + threadIdx.x;        // destination entry
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  long long r = blk;
  const int m = (int)(r % WM); r /= WM;
  const int n = (int)(r & 1); r >>= 1;
  const int wave = (int)(r & 3); r >>= 2;
  const int tile = (int)r;
  const int b = tile / (tiles_x * tiles_y), trem = tile - b * tiles_x * tiles_y;
  const int ty = trem / tiles_x, tx = trem - ty * tiles_x;
  const int y = ty * TH + wave * WM + m;
  (void)B;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (y < stiles_y * 8) {
    const int sty = y >> 3, swave = (y & 7) >> 1, sm = y & 1;
    const long long stile = ((long long)b * stiles_y + sty) * tiles_x + tx;
    v = src[((((stile * 4 + swave) * 2 + n) * 2 + sm) * 4 + q) * 64 + lane];
  }
  if (out_kind == 2) {
    // NaN-preserving maximum (fmaxf drops a NaN operand and v_cvt_pknorm_i16_f32(NaN) is 0: a NaN in the condition map would come out of the
    // hoisted term as finite zeros -- the other modes propagate it).  A NaN anywhere in the block makes the block's SCALE NaN, and the
    // consumer's int16 * scale reproduces it on the block's 32 pixels x 32 couts.
    auto nmax = [](float a, float b) { return (b > a || b != b) ? b : a; };
    float mx = nmax(nmax(fabsf(v.x), fabsf(v.y)), nmax(fabsf(v.z), fabsf(v.w)));
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = nmax(mx, __shfl_xor(mx, off, 64));
    if (lane == 0) s_max[q] = mx;
    __syncthreads();
    mx = nmax(nmax(s_max[0], s_max[1]), nmax(s_max[2], s_max[3]));
    const float inv = mx > 0.f ? 1.f / mx : 0.f;
    if (threadIdx.x == 0) scales[blk] = mx * (1.f / Q15_ONE);
    reinterpret_cast<uint2*>(dst)[e] = make_uint2(DD_CVT_PKNORM_I16(v.x * inv, v.y * inv), DD_CVT_PKNORM_I16(v.z * inv, v.w * inv));
  } else reinterpret_cast<uint2*>(dst)[e] = make_uint2(pack2<EK_F16>(v.x, v.y), pack2<EK_F16>(v.z, v.w));
}
hipError_t launch_cadd_reformat(const float* src, void* dst, float* scales, int B, int h, int w, int big, int out_kind, hipStream_t s) {
  const int TH = big ? 16 : 8, WM = big ? 4 : 2;
  const long long n_blocks = (long long)B * ((h + TH - 1) / TH) * ((w + 31) / 32) * 4 * 2 * WM;
  hipLaunchKernelGGL(cadd_reformat_kernel, dim3((unsigned)n_blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(src), dst, scales, B, h, w, big, out_kind);
  return hipGetLastError();
}

// W[co][ci][k] -> W'[ci][co][kk-1-k] (fp32; the naive backward's data-gradient weights)
__global__ void __launch_bounds__(256) transpose_flip_kernel(const float* __restrict__ w, float* __restrict__ wt, int cout, int cin, int kk) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)cout * cin * kk) return;
  const int k = (int)(i % kk), ci = (int)((i / kk) % cin), co = (int)(i / ((long long)kk * cin));
  wt[((size_t)ci * cout + co) * kk + (kk - 1 - k)] = w[i];
}
hipError_t launch_transpose_flip(const float* w, float* wt, int cout, int cin, int kk, hipStream_t s) {
  const long long n = (long long)cout * cin * kk;
  hipLaunchKernelGGL(transpose_flip_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, wt, cout, cin, kk);
  return hipGetLastError();
}

}  // namespace dd
