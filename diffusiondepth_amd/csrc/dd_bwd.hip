// dd_bwd.hip -- backward (vector-Jacobian product) pieces of one epsilon-network evaluation, SURVEY.md 8f rank 2:
// what autograd does for  eps = ScheduledCNNRefine.forward(x_t, t, cond)  (reference
// src/model/head/ddim_depth_estimate_res.py:324-344) when the reference trains (loss.backward(), src/main.py:232-241).
//
// Per layer L (conv -> GroupNorm(4) -> ReLU), with y = conv output, yh = (y - mu_g) * rstd_g, z = gamma*yh + beta, a = relu(z)
// and g_a = dLoss/da arriving from the layer behind it:
//     g_z      = g_a * [z > 0]
//     dbeta_c  = sum_{b,p} g_z          dgamma_c = sum_{b,p} g_z * yh                              (gn_bwd_reduce)
//     g_y      = rstd_g * ( gamma_c g_z - mean_g(gamma g_z) - yh * mean_g(gamma g_z yh) )          (gn_bwd_apply)
//     dW       = sum_{b,p} g_y (x) a_prev(patch),  dbias = sum g_y                                 (wgrad, channel_sum)
//     g_a_prev = conv3x3(g_y, W^T flipped)                                                         (dgrad: the forward conv kernels)
// The elementwise / reduction kernels here address activations through ActView, so one source serves the unfused fp32
// cross-check path (plain NHWC fp32) and the fused path's channel-blocked bf16 / f16 / fp32 tensors.
#include <cmath>
#include "dd_elem.h"

namespace dd {

__device__ __forceinline__ float view_load(const ActView& v, long long b, long long p, int c) {
  const size_t idx = (v.blocked && v.C >= ACT_CB) ? (((size_t)b * (v.C / ACT_CB) + c / ACT_CB) * v.HW + p) * ACT_CB + (c % ACT_CB)
                                                  : ((size_t)b * v.HW + p) * v.C + c;
  if (v.ek == EK_F32) return reinterpret_cast<const float*>(v.p)[idx];
  const uint32_t u = reinterpret_cast<const uint16_t*>(v.p)[idx];
  return v.ek == EK_BF16 ? __builtin_bit_cast(float, u << 16) : (float)__builtin_bit_cast(_Float16, (uint16_t)u);
}
__device__ __forceinline__ void view_store(const ActView& v, long long b, long long p, int c, float f) {
  const size_t idx = (v.blocked && v.C >= ACT_CB) ? (((size_t)b * (v.C / ACT_CB) + c / ACT_CB) * v.HW + p) * ACT_CB + (c % ACT_CB)
                                                  : ((size_t)b * v.HW + p) * v.C + c;
  void* q = const_cast<void*>(v.p);
  if (v.ek == EK_F32) reinterpret_cast<float*>(q)[idx] = f;
  else if (v.ek == EK_BF16) reinterpret_cast<uint16_t*>(q)[idx] = (uint16_t)f32_to_bf16(f);
  else reinterpret_cast<uint16_t*>(q)[idx] = (uint16_t)f32_to_f16(f);
}

// mean / rstd of group grp of sample b from the forward pass's fp64 partial sums (same arithmetic as the forward prologues)
__device__ __forceinline__ void group_moments(const double* __restrict__ stats, long long b, int grp, double cnt, float& mean, float& rstd) {
  const double* st = stats + (size_t)b * STAT_SLOTS * STAT_STRIDE;
  double s = 0.0, q = 0.0;
  for (int k = 0; k < STAT_SLOTS; ++k) { s += st[k * STAT_STRIDE + grp * 2]; q += st[k * STAT_STRIDE + grp * 2 + 1]; }
  const double m = s / cnt;
  double var = q / cnt - m * m;
  var = var > 0.0 ? var : 0.0;
  mean = (float)m;
  rstd = (float)(1.0 / sqrt(var + (double)GN_EPS));
}

// ------------------------------------------------------------------------------------------------
// per (sample, channel):  out[b][c][0] += sum_p g_a * [z > 0],   out[b][c][1] += sum_p g_a * [z > 0] * yh     (fp64)
// grid (pixel slabs, B); 256 threads = (256 / C) pixel lanes x C channels (C = 16 / 64 / 256).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gn_bwd_reduce_kernel(ActView ga, ActView y, const double* __restrict__ stats,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            double* __restrict__ out, int slab) {
  __shared__ float s_mean[GN_GROUPS], s_rstd[GN_GROUPS];
  __shared__ double s_red[256][2];
  const int C = y.C;
  const long long b = blockIdx.y;
  const int tid = threadIdx.x;
  if (tid < GN_GROUPS) group_moments(stats, b, tid, (double)y.HW * (C / GN_GROUPS), s_mean[tid], s_rstd[tid]);
  __syncthreads();
  const int c = tid % C, lane_p = tid / C, np = 256 / C;
  const int grp = c / (C / GN_GROUPS);
  const float mean = s_mean[grp], rstd = s_rstd[grp], gm = gamma[c], bt = beta[c];
  const long long p0 = (long long)blockIdx.x * slab, p1 = min(p0 + slab, y.HW);
  double sb = 0.0, sg = 0.0;
  float fb = 0.f, fg = 0.f;
  int n = 0;
  for (long long p = p0 + lane_p; p < p1; p += np) {
    const float yh = (view_load(y, b, p, c) - mean) * rstd;
    const float z = fmaf(gm, yh, bt);
    const float g = z > 0.f ? view_load(ga, b, p, c) : 0.f;
    fb += g; fg = fmaf(g, yh, fg);
    if (++n == 64) { sb += fb; sg += fg; fb = 0.f; fg = 0.f; n = 0; }     // fp32 runs of <= 64 values, fp64 above
  }
  sb += fb; sg += fg;
  s_red[tid][0] = sb; s_red[tid][1] = sg;
  __syncthreads();
  if (tid < C) {
    double tb = 0.0, tg = 0.0;
    for (int k = 0; k < np; ++k) { tb += s_red[k * C + tid][0]; tg += s_red[k * C + tid][1]; }
    double* dst = out + ((size_t)b * C + tid) * 2;
    atomicAdd(dst, tb);
    atomicAdd(dst + 1, tg);
  }
}
hipError_t launch_gn_bwd_reduce(const ActView& ga, const ActView& y, const double* stats, const float* gamma, const float* beta,
                                double* out_bc2, int B, hipStream_t s) {
  if (256 % y.C != 0) return hipErrorInvalidValue;
  const int slab = (y.C <= 16) ? 256 : 2048;            // 16-channel tensors: 16 pixel lanes per block, keep the grid large
  dim3 grid((unsigned)((y.HW + slab - 1) / slab), (unsigned)B);
  hipLaunchKernelGGL(gn_bwd_reduce_kernel, grid, dim3(256), 0, s, ga, y, stats, gamma, beta, out_bc2, slab);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// g_y = rstd * (gamma g_z - S1/N - yh S2/N),  S1 = sum_{c in g} gamma_c dbeta_bc,  S2 = sum gamma_c dgamma_bc,
// N = (C/4) h w.  Optionally also materialises the layer's activation  act = relu(z) [+ cond + E[t]]  (the next conv's
// input, which wgrad contracts with).  grid (element blocks, B).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gn_bwd_apply_kernel(ActView ga, ActView y, const double* __restrict__ stats,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const double* __restrict__ dgb, ActView gy, ActView act, ActView cond,
                                                           const float* __restrict__ emb, const long long* __restrict__ tvec,
                                                           int t_base, int t_bstride) {
  __shared__ float s_mean[GN_GROUPS], s_rstd[GN_GROUPS], s_s1[GN_GROUPS], s_s2[GN_GROUPS];
  const int C = y.C, CG = C / GN_GROUPS;
  const long long b = blockIdx.y;
  const int tid = threadIdx.x;
  if (tid < GN_GROUPS) {
    group_moments(stats, b, tid, (double)y.HW * CG, s_mean[tid], s_rstd[tid]);
    double a1 = 0.0, a2 = 0.0;
    for (int c = tid * CG; c < (tid + 1) * CG; ++c) {
      a1 += (double)gamma[c] * dgb[((size_t)b * C + c) * 2];
      a2 += (double)gamma[c] * dgb[((size_t)b * C + c) * 2 + 1];
    }
    const double inv_n = 1.0 / ((double)y.HW * CG);
    s_s1[tid] = (float)(a1 * inv_n);
    s_s2[tid] = (float)(a2 * inv_n);
  }
  __syncthreads();
  const long long total = y.HW * C;
  const long long t = tvec ? clamp_t(tvec[t_base + b * t_bstride]) : 0;
  for (long long i = (long long)blockIdx.x * 256 + tid; i < total; i += (long long)gridDim.x * 256) {
    const long long p = i / C;
    const int c = (int)(i - p * C);
    const int grp = c / CG;
    const float yh = (view_load(y, b, p, c) - s_mean[grp]) * s_rstd[grp];
    const float z = fmaf(gamma[c], yh, beta[c]);
    if (gy.p) {
      const float g = z > 0.f ? view_load(ga, b, p, c) : 0.f;
      view_store(gy, b, p, c, s_rstd[grp] * (gamma[c] * g - s_s1[grp] - yh * s_s2[grp]));
    }
    if (act.p) {
      float a = fmaxf(z, 0.f);
      if (cond.p) a = (view_load(cond, b, p, c) + emb[(size_t)t * C + c]) + a;
      view_store(act, b, p, c, a);
    }
  }
}
hipError_t launch_gn_bwd_apply(const ActView& ga, const ActView& y, const double* stats, const float* gamma, const float* beta,
                               const double* dgb, const ActView& gy, const ActView& act, const ActView& cond, const float* emb,
                               const long long* tvec, int t_base, int t_bstride, int B, hipStream_t s) {
  const long long total = y.HW * y.C;
  const unsigned nb = (unsigned)min((long long)4096, (total + 255) / 256);
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(nb, (unsigned)B), dim3(256), 0, s, ga, y, stats, gamma, beta, dgb, gy, act, cond, emb,
                     tvec, t_base, t_bstride);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// out[row(b)][c] += sum_p v[b][p][c]  (fp32 atomics of per-block fp64 sums): conv bias gradients (row = 0 for every b) and
// the time-embedding gradient (row = t_b).  grid (pixel slabs, B).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) channel_sum_kernel(ActView v, float* __restrict__ out, const long long* __restrict__ rows,
                                                          int t_base, int t_bstride, int slab) {
  __shared__ double s_red[256];
  const int C = v.C;
  const long long b = blockIdx.y;
  const int tid = threadIdx.x;
  const int c = tid % C, lane_p = tid / C, np = 256 / C;
  const long long p0 = (long long)blockIdx.x * slab, p1 = min(p0 + slab, v.HW);
  double acc = 0.0;
  float f = 0.f;
  int n = 0;
  for (long long p = p0 + lane_p; p < p1; p += np) {
    f += view_load(v, b, p, c);
    if (++n == 64) { acc += f; f = 0.f; n = 0; }
  }
  acc += f;
  s_red[tid] = acc;
  __syncthreads();
  if (tid < C) {
    double t = 0.0;
    for (int k = 0; k < np; ++k) t += s_red[k * C + tid];
    const long long row = rows ? clamp_t(rows[t_base + b * t_bstride]) : 0;
    atomicAdd(out + (size_t)row * C + tid, (float)t);
  }
}
hipError_t launch_channel_sum(const ActView& v, float* out, const long long* rows, int t_base, int t_bstride, int B, hipStream_t s) {
  if (256 % v.C != 0) return hipErrorInvalidValue;
  const int slab = (v.C <= 16) ? 256 : 2048;
  dim3 grid((unsigned)((v.HW + slab - 1) / slab), (unsigned)B);
  hipLaunchKernelGGL(channel_sum_kernel, grid, dim3(256), 0, s, v, out, rows, t_base, t_bstride, slab);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Vectorised forms for the fused path's channel-blocked 2-byte tensors (C = 64 / 256): a thread owns one 16-byte piece
// (8 channels) of a pixel, 4 threads cover a 32-channel block of that pixel, consecutive lanes = consecutive addresses.
// grid (pixel slabs, C / 32, B), 256 threads = 64 pixels x 4 pieces per pass.
// ------------------------------------------------------------------------------------------------
template <int EK>
__device__ __forceinline__ void unpack8(const uint4& v, float (&f)[8]) { Piece<EK>::unpack(v, f); }
// eight consecutive channels of a channel-blocked tensor of kind K at ELEMENT offset `off`: one 16-byte piece of a 2-byte kind, two of fp32 (the stored
// conv outputs and the condition map of the split-f16 mode, whose gradients travel as f16: DD_PREC_F16X3's backward)
template <int K> struct Raw8 {
  uint4 v[K == EK_F32 ? 2 : 1];
  __device__ __forceinline__ void load(const void* p, size_t off) {
    if constexpr (K == EK_F32) {
      const uint4* q = reinterpret_cast<const uint4*>(static_cast<const float*>(p) + off);
      v[0] = q[0]; v[1] = q[1];
    } else {
      v[0] = *reinterpret_cast<const uint4*>(static_cast<const uint16_t*>(p) + off);
    }
  }
  __device__ __forceinline__ void unpack(float (&f)[8]) const {
    if constexpr (K == EK_F32) {
      float a[4], b[4];
      Piece<EK_F32>::unpack(v[0], a); Piece<EK_F32>::unpack(v[1], b);
#pragma unroll
      for (int i = 0; i < 4; ++i) { f[i] = a[i]; f[4 + i] = b[i]; }
    } else {
      Piece<K>::unpack(v[0], f);
    }
  }
};

// per (b, c):  out[0] += sum g_z,  out[1] += sum g_z * yh,  out[2] += sum y  (for the analytic conv-bias gradient),
// out[3] += sum g_a (unmasked: the time-embedding gradient when g_a is dLoss/df)
// EK = kind of the gradient tensor, YK = kind of the stored conv output (they differ in the mode EK_BF16M: bf16 gradients, f16 y)
template <int EK, int YK>
__global__ void __launch_bounds__(256) gn_bwd_reduce_blocked_kernel(const uint16_t* __restrict__ ga, const void* __restrict__ y,
                                                                    const double* __restrict__ stats, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, double* __restrict__ out, int C,
                                                                    long long HW, int slab) {
  __shared__ float s_mean[GN_GROUPS], s_rstd[GN_GROUPS];
  __shared__ float s_red[64][4][33];                       // [pixel lane][sum kind][channel of the block] (+1 pad)
  const int tid = threadIdx.x;
  const long long b = blockIdx.z;
  const int cb = blockIdx.y;
  if (tid < GN_GROUPS) group_moments(stats, b, tid, (double)HW * (C / GN_GROUPS), s_mean[tid], s_rstd[tid]);
  __syncthreads();
  const int q = tid & 3, pl = tid >> 2;
  const int c0 = cb * ACT_CB + q * 8;
  const int grp = c0 / (C / GN_GROUPS);                    // 8 channels never straddle a group (C/4 >= 16)
  const float mean = s_mean[grp], rstd = s_rstd[grp];
  float gm[8], bt[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { gm[k] = gamma[c0 + k]; bt[k] = beta[c0 + k]; }
  float a0[8], a1[8], a2[8], a3[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { a0[k] = 0.f; a1[k] = 0.f; a2[k] = 0.f; a3[k] = 0.f; }
  const size_t base = ((size_t)b * (C / ACT_CB) + cb) * HW * ACT_CB + q * 8;
  const long long p0 = (long long)blockIdx.x * slab, p1 = min(p0 + slab, HW);
  for (long long p = p0 + pl; p < p1; p += 64) {
    const uint4 gv = *reinterpret_cast<const uint4*>(ga + base + (size_t)p * ACT_CB);
    Raw8<YK> yv; yv.load(y, base + (size_t)p * ACT_CB);
    float g[8], yy[8];
    unpack8<EK>(gv, g); yv.unpack(yy);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float yh = (yy[k] - mean) * rstd;
      const float gz = fmaf(gm[k], yh, bt[k]) > 0.f ? g[k] : 0.f;
      a0[k] += gz; a1[k] = fmaf(gz, yh, a1[k]); a2[k] += yy[k]; a3[k] += g[k];
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { s_red[pl][0][q * 8 + k] = a0[k]; s_red[pl][1][q * 8 + k] = a1[k]; s_red[pl][2][q * 8 + k] = a2[k]; s_red[pl][3][q * 8 + k] = a3[k]; }
  __syncthreads();
  if (tid < 128) {                                         // (kind, channel): sum the 64 pixel lanes in fp64
    const int kind = tid >> 5, c = tid & 31;
    double t = 0.0;
    for (int l = 0; l < 64; ++l) t += (double)s_red[l][kind][c];
    atomicAdd(out + ((size_t)b * C + cb * ACT_CB + c) * 4 + kind, t);
  }
}
hipError_t launch_gn_bwd_reduce_blocked(const void* ga, const void* y, int ek, int yk, const double* stats, const float* gamma, const float* beta,
                                        double* out_bc4, int B, int C, long long HW, hipStream_t s) {
  if (C % ACT_CB != 0 || (ek != EK_BF16 && ek != EK_F16) || !(yk == ek || (ek == EK_BF16 && yk == EK_F16) || (ek == EK_F16 && yk == EK_F32))) return hipErrorInvalidValue;
  const int slab = 1024;                                   // 16 passes of 64 pixels: fp32 partials of <= 16 values per thread
  dim3 grid((unsigned)((HW + slab - 1) / slab), (unsigned)(C / ACT_CB), (unsigned)B);
  if (ek == EK_BF16 && yk == EK_F16) hipLaunchKernelGGL((gn_bwd_reduce_blocked_kernel<EK_BF16, EK_F16>), grid, dim3(256), 0, s, reinterpret_cast<const uint16_t*>(ga),
                                        y, stats, gamma, beta, out_bc4, C, HW, slab);
  else if (ek == EK_BF16) hipLaunchKernelGGL((gn_bwd_reduce_blocked_kernel<EK_BF16, EK_BF16>), grid, dim3(256), 0, s, reinterpret_cast<const uint16_t*>(ga),
                                        y, stats, gamma, beta, out_bc4, C, HW, slab);
  else if (yk == EK_F32) hipLaunchKernelGGL((gn_bwd_reduce_blocked_kernel<EK_F16, EK_F32>), grid, dim3(256), 0, s, reinterpret_cast<const uint16_t*>(ga),
                                        y, stats, gamma, beta, out_bc4, C, HW, slab);      // split-f16 mode: fp32 y, f16 gradients
  else hipLaunchKernelGGL((gn_bwd_reduce_blocked_kernel<EK_F16, EK_F16>), grid, dim3(256), 0, s, reinterpret_cast<const uint16_t*>(ga),
                          y, stats, gamma, beta, out_bc4, C, HW, slab);
  return hipGetLastError();
}

// g_y = A_c g_z + P_g + Q_g y   with A_c = gamma_c rstd_g,  Q_g = -rstd^2 S2/N,  P_g = -rstd S1/N - mean Q_g  (the same
// expression as gn_bwd_apply_kernel, regrouped);  optional  act = relu(A_c y + B_c) [+ cond + E[t]].   sums: [b][c][4].
template <int EK, int YK>      // EK: g_a, g_y and act;  YK: y and cond
__global__ void __launch_bounds__(256) gn_bwd_apply_blocked_kernel(const uint16_t* __restrict__ ga, const void* __restrict__ y,
                                                                   const double* __restrict__ stats, const float* __restrict__ gamma,
                                                                   const float* __restrict__ beta, const double* __restrict__ sums,
                                                                   uint16_t* __restrict__ gy, uint16_t* __restrict__ act,
                                                                   const void* __restrict__ cond, const float* __restrict__ emb,
                                                                   const long long* __restrict__ tvec, int t_base, int t_bstride, int C,
                                                                   long long HW, int slab) {
  __shared__ float s_mean[GN_GROUPS], s_rstd[GN_GROUPS], s_p[GN_GROUPS], s_q[GN_GROUPS];
  const int tid = threadIdx.x;
  const long long b = blockIdx.z;
  const int cb = blockIdx.y;
  const int CG = C / GN_GROUPS;
  if (tid < GN_GROUPS) {
    float mean, rstd;
    group_moments(stats, b, tid, (double)HW * CG, mean, rstd);
    double s1 = 0.0, s2 = 0.0;
    if (gy) {
      for (int c = tid * CG; c < (tid + 1) * CG; ++c) {
        s1 += (double)gamma[c] * sums[((size_t)b * C + c) * 4];
        s2 += (double)gamma[c] * sums[((size_t)b * C + c) * 4 + 1];
      }
    }
    const double inv_n = 1.0 / ((double)HW * CG);
    const float qg = -(rstd * rstd) * (float)(s2 * inv_n);
    s_mean[tid] = mean; s_rstd[tid] = rstd;
    s_q[tid] = qg;
    s_p[tid] = -rstd * (float)(s1 * inv_n) - mean * qg;
  }
  __syncthreads();
  const int q = tid & 3, pl = tid >> 2;
  const int c0 = cb * ACT_CB + q * 8;
  const int grp = c0 / CG;
  const float mean = s_mean[grp], rstd = s_rstd[grp], pg = s_p[grp], qg = s_q[grp];
  float ta[8], tb[8], te[8];
  const long long t = tvec ? clamp_t(tvec[t_base + b * t_bstride]) : 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    ta[k] = gamma[c0 + k] * rstd;
    tb[k] = beta[c0 + k] - ta[k] * mean;
    te[k] = (cond && emb) ? emb[(size_t)t * C + c0 + k] : 0.f;
  }
  const size_t base = ((size_t)b * (C / ACT_CB) + cb) * HW * ACT_CB + q * 8;
  const long long p0 = (long long)blockIdx.x * slab, p1 = min(p0 + slab, HW);
  // two pixels per trip: every load of both is issued before the first result is needed (the kernel streams 3-4 tensors and is bound by
  // the bytes in flight: 12 waves per CU x one 16-B load per tensor reached 46 % of the HBM rate)
  for (long long p = p0 + pl; p < p1; p += 128) {
    const bool two = p + 64 < p1;
    const size_t off[2] = {base + (size_t)p * ACT_CB, base + (size_t)(two ? p + 64 : p) * ACT_CB};
    Raw8<YK> ry[2], rc[2];
    uint4 rg[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      ry[j].load(y, off[j]);
      if (gy) rg[j] = *reinterpret_cast<const uint4*>(ga + off[j]);
      if (act && cond) rc[j].load(cond, off[j]);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      if (j == 1 && !two) break;
      float yy[8];
      ry[j].unpack(yy);
      if (gy) {
        float g[8], o[8];
        unpack8<EK>(rg[j], g);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float gz = fmaf(ta[k], yy[k], tb[k]) > 0.f ? g[k] : 0.f;
          o[k] = fmaf(ta[k], gz, fmaf(qg, yy[k], pg));
        }
        *reinterpret_cast<uint4*>(gy + off[j]) = Piece<EK>::pack(o);
      }
      if (act) {
        float o[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[k] = fmaxf(fmaf(ta[k], yy[k], tb[k]), 0.f);
        if (cond) {
          float cv[8];
          rc[j].unpack(cv);
#pragma unroll
          for (int k = 0; k < 8; ++k) o[k] = o[k] + (cv[k] + te[k]);
        }
        *reinterpret_cast<uint4*>(act + off[j]) = Piece<EK>::pack(o);
      }
    }
  }
}
hipError_t launch_gn_bwd_apply_blocked(const void* ga, const void* y, int ek, int yk, const double* stats, const float* gamma, const float* beta,
                                       const double* sums_bc4, void* gy, void* act, const void* cond, const float* emb,
                                       const long long* tvec, int t_base, int t_bstride, int B, int C, long long HW, hipStream_t s) {
  if (C % ACT_CB != 0 || (ek != EK_BF16 && ek != EK_F16) || !(yk == ek || (ek == EK_BF16 && yk == EK_F16) || (ek == EK_F16 && yk == EK_F32))) return hipErrorInvalidValue;
  const int slab = 2048;      // the per-workgroup prologue (group moments, gamma-weighted sums over C/4 channels in fp64) costs as much as 512 pixels of streaming
  dim3 grid((unsigned)((HW + slab - 1) / slab), (unsigned)(C / ACT_CB), (unsigned)B);
  auto U16 = [](const void* p) { return reinterpret_cast<const uint16_t*>(p); };
  if (ek == EK_BF16 && yk == EK_F16) hipLaunchKernelGGL((gn_bwd_apply_blocked_kernel<EK_BF16, EK_F16>), grid, dim3(256), 0, s, U16(ga), y, stats, gamma, beta, sums_bc4,
                                        reinterpret_cast<uint16_t*>(gy), reinterpret_cast<uint16_t*>(act), cond, emb, tvec, t_base, t_bstride, C, HW, slab);
  else if (ek == EK_BF16) hipLaunchKernelGGL((gn_bwd_apply_blocked_kernel<EK_BF16, EK_BF16>), grid, dim3(256), 0, s, U16(ga), y, stats, gamma, beta, sums_bc4,
                                        reinterpret_cast<uint16_t*>(gy), reinterpret_cast<uint16_t*>(act), cond, emb, tvec, t_base, t_bstride, C, HW, slab);
  else if (yk == EK_F32) hipLaunchKernelGGL((gn_bwd_apply_blocked_kernel<EK_F16, EK_F32>), grid, dim3(256), 0, s, U16(ga), y, stats, gamma, beta, sums_bc4,
                                        reinterpret_cast<uint16_t*>(gy), reinterpret_cast<uint16_t*>(act), cond, emb, tvec, t_base, t_bstride, C, HW, slab);
  else hipLaunchKernelGGL((gn_bwd_apply_blocked_kernel<EK_F16, EK_F16>), grid, dim3(256), 0, s, U16(ga), y, stats, gamma, beta, sums_bc4,
                          reinterpret_cast<uint16_t*>(gy), reinterpret_cast<uint16_t*>(act), cond, emb, tvec, t_base, t_bstride, C, HW, slab);
  return hipGetLastError();
}

// Parameter gradients of one layer from the per (b, c) sums [B][C][4] of gn_bwd_reduce_blocked_kernel:
//   dbeta_c += sum_b s0,  dgamma_c += sum_b s1,  dbias_c += sum_b sum_p g_y = sum_b (A_c s0 + HW P_g + Q_g s2)   (g_y as above),
//   demb[t_b][c] += s3 (only when demb != NULL).
// One workgroup per GroupNorm group, one thread per channel of it: the gamma-weighted group sums S1, S2 of every image are reduced across the
// group's threads in LDS (each thread used to re-read all C/4 channels' sums per image: 32 us of dependent loads for a kernel that moves 16 KB).
__global__ void gn_param_grad4_kernel(const double* __restrict__ sums, const double* __restrict__ stats, const float* __restrict__ gamma,
                                      float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dbias, float* __restrict__ demb,
                                      const long long* __restrict__ tvec, int t_base, int t_bstride, int B, int C, long long HW) {
  __shared__ double s_part[2][64];
  __shared__ double s_tot[2];
  const int CG = C / GN_GROUPS, grp = blockIdx.x, tid = threadIdx.x;      // blockDim.x == CG <= 64
  const int c = grp * CG + tid;
  const double g = (double)gamma[c];
  double sb = 0.0, sg = 0.0, sbias = 0.0;
  for (int b = 0; b < B; ++b) {
    const double* sc = sums + ((size_t)b * C + c) * 4;
    const double s0 = sc[0], s1c = sc[1], s2c = sc[2], s3c = sc[3];
    s_part[0][tid] = g * s0;
    s_part[1][tid] = g * s1c;
    __syncthreads();
    if (tid < 2) {
      double t = 0.0;
      for (int k = 0; k < CG; ++k) t += s_part[tid][k];      // channel order: the sums of the previous kernel, bit for bit
      s_tot[tid] = t;
    }
    __syncthreads();
    const double s1 = s_tot[0], s2 = s_tot[1];
    float mean, rstd;
    group_moments(stats, b, grp, (double)HW * CG, mean, rstd);
    const double inv_n = 1.0 / ((double)HW * CG);
    const double qg = -(double)rstd * rstd * s2 * inv_n;
    const double pg = -(double)rstd * s1 * inv_n - (double)mean * qg;
    sb += s0; sg += s1c;
    sbias += g * rstd * s0 + (double)HW * pg + qg * s2c;
    if (demb) atomicAdd(demb + (size_t)clamp_t(tvec[t_base + b * t_bstride]) * C + c, (float)s3c);
    __syncthreads();                                          // s_part / s_tot are rewritten for the next image
  }
  dbeta[c] += (float)sb;
  dgamma[c] += (float)sg;
  dbias[c] += (float)sbias;
}
hipError_t launch_gn_param_grad4(const double* sums_bc4, const double* stats, const float* gamma, float* dgamma, float* dbeta, float* dbias,
                                 float* demb, const long long* tvec, int t_base, int t_bstride, int B, int C, long long HW, hipStream_t s) {
  if (C % GN_GROUPS != 0 || C / GN_GROUPS > 64) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gn_param_grad4_kernel, dim3(GN_GROUPS), dim3(C / GN_GROUPS), 0, s, sums_bc4, stats, gamma, dgamma, dbeta, dbias, demb, tvec,
                     t_base, t_bstride, B, C, HW);
  return hipGetLastError();
}

// out[c] += sum_{b,p} v[b][p][c] for a channel-blocked 2-byte tensor (conv bias gradients of the Swin fuse convs, which have
// no GroupNorm behind them).  Same thread mapping as gn_bwd_reduce_blocked_kernel.
template <int EK>
__global__ void __launch_bounds__(256) channel_sum_blocked_kernel(const uint16_t* __restrict__ v, float* __restrict__ out, int C, long long HW, int slab) {
  __shared__ float s_red[64][33];
  const int tid = threadIdx.x;
  const long long b = blockIdx.z;
  const int cb = blockIdx.y, q = tid & 3, pl = tid >> 2;
  float a[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) a[k] = 0.f;
  const size_t base = ((size_t)b * (C / ACT_CB) + cb) * HW * ACT_CB + q * 8;
  const long long p0 = (long long)blockIdx.x * slab, p1 = min(p0 + slab, HW);
  for (long long p = p0 + pl; p < p1; p += 64) {
    float f[8];
    unpack8<EK>(*reinterpret_cast<const uint4*>(v + base + (size_t)p * ACT_CB), f);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] += f[k];
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) s_red[pl][q * 8 + k] = a[k];
  __syncthreads();
  if (tid < 32) {
    double t = 0.0;
    for (int l = 0; l < 64; ++l) t += (double)s_red[l][tid];
    atomicAdd(out + cb * ACT_CB + tid, (float)t);
  }
}
hipError_t launch_channel_sum_blocked(const void* v, int ek, float* out, int B, int C, long long HW, hipStream_t s) {
  if (C % ACT_CB != 0 || (ek != EK_BF16 && ek != EK_F16)) return hipErrorInvalidValue;
  const int slab = 1024;
  dim3 grid((unsigned)((HW + slab - 1) / slab), (unsigned)(C / ACT_CB), (unsigned)B);
  if (ek == EK_BF16) hipLaunchKernelGGL(channel_sum_blocked_kernel<EK_BF16>, grid, dim3(256), 0, s, reinterpret_cast<const uint16_t*>(v), out, C, HW, slab);
  else hipLaunchKernelGGL(channel_sum_blocked_kernel<EK_F16>, grid, dim3(256), 0, s, reinterpret_cast<const uint16_t*>(v), out, C, HW, slab);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Adjoint of the bilinear upsample (align_corners=True) of the Swin condition map: dst[b][c][sy][sx] (NCHW fp32, (ch, cw))
// (+)= sum over the latent pixels (oy, ox) whose interpolation stencil contains (sy, sx) of weight * g[b][oy][ox][c].
// Gather form (no atomics): the candidate rows / columns of a source pixel are an interval, the weights are recomputed with
// exactly the forward's fp32 expressions (upsample_to_blocked_kernel).  One thread per (source pixel, 8-channel piece).
// ------------------------------------------------------------------------------------------------
template <int EK>
__global__ void __launch_bounds__(256) upsample_adjoint_kernel(const uint16_t* __restrict__ g, float* __restrict__ dst, int C, int ch, int cw,
                                                              int h, int w, int accumulate, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int npc = C / 8;
  const int piece = (int)(i % npc);
  long long r = i / npc;
  const int sx = (int)(r % cw); r /= cw;
  const int sy = (int)(r % ch);
  const long long b = r / ch;
  const float fsy = (h > 1) ? (float)(ch - 1) / (float)(h - 1) : 0.f;
  const float fsx = (w > 1) ? (float)(cw - 1) / (float)(w - 1) : 0.f;
  // destination rows whose y0 or y1 can equal sy: fy = fsy * oy in (sy - 1, sy + 1)
  const int oy_lo = (fsy > 0.f) ? max(0, (int)floorf((float)(sy - 1) / fsy)) : 0;
  const int oy_hi = (fsy > 0.f) ? min(h - 1, (int)ceilf((float)(sy + 1) / fsy)) : h - 1;
  const int ox_lo = (fsx > 0.f) ? max(0, (int)floorf((float)(sx - 1) / fsx)) : 0;
  const int ox_hi = (fsx > 0.f) ? min(w - 1, (int)ceilf((float)(sx + 1) / fsx)) : w - 1;
  const int c0 = piece * 8;
  const size_t gbase = ((size_t)b * (C / ACT_CB) + c0 / ACT_CB) * (size_t)h * w * ACT_CB + (c0 % ACT_CB);
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  for (int oy = oy_lo; oy <= oy_hi; ++oy) {
    const float fy = fsy * (float)oy;
    const int y0 = min((int)fy, ch - 1), y1 = min(y0 + 1, ch - 1);
    const float ly = fy - (float)y0, hy = 1.f - ly;
    const float wy = (y0 == sy ? hy : 0.f) + (y1 == sy ? ly : 0.f);
    if (wy == 0.f) continue;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
      const float fx = fsx * (float)ox;
      const int x0 = min((int)fx, cw - 1), x1 = min(x0 + 1, cw - 1);
      const float lx = fx - (float)x0, hx = 1.f - lx;
      const float wx = (x0 == sx ? hx : 0.f) + (x1 == sx ? lx : 0.f);
      if (wx == 0.f) continue;
      float f[8];
      unpack8<EK>(*reinterpret_cast<const uint4*>(g + gbase + ((size_t)oy * w + ox) * ACT_CB), f);
      const float wt = wy * wx;
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = fmaf(wt, f[k], acc[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float* o = dst + (((size_t)b * C + c0 + k) * ch + sy) * cw + sx;
    *o = accumulate ? *o + acc[k] : acc[k];
  }
}
// Tiled separable form of the same sum for the 16-bit channel-blocked gradient: one workgroup per (image, 32-channel block, source row,
// segment of ADJ_SXT source columns).  Pass 1 (vertical): thread (destination column, 16-B piece) sums wy * g over the destination rows
// whose stencil touches source row sy -- consecutive lanes read consecutive 16-B pieces of ONE destination row (whole 64-B channel-block
// lines; the lanes of upsample_adjoint_kernel sit h*w*64 B apart and every g element is fetched ~4 times in 64-B requests) -- and leaves
// the fp32 row in LDS.  Pass 2 (horizontal): thread (source column, channel) sums wx * row; 16 consecutive floats per channel go to the
// NCHW gradient.  The weights are the forward's own fp32 expressions; only the summation order differs from the kernel above.
constexpr int ADJ_SXT = 64;        // source columns per workgroup
constexpr int ADJ_MAXOX = 288;     // destination columns a segment may touch (scale factors up to ~4; beyond: the kernel above)
constexpr int ADJ_MAXR = 20;       // destination rows with a non-zero weight for one source row (2 x scale + 1)
template <int EK>
__global__ void __launch_bounds__(256) upsample_adjoint_tiled_kernel(const uint16_t* __restrict__ g, float* __restrict__ dst, int C, int ch, int cw,
                                                                    int h, int w, int accumulate, int n_seg) {
  __shared__ float tmp[ADJ_MAXOX * 33];
  const int tid = threadIdx.x;
  int r = blockIdx.x;
  const int seg = r % n_seg; r /= n_seg;
  const int sy = r % ch; r /= ch;
  const int ncb = C / ACT_CB;
  const int cb = r % ncb;
  const int b = r / ncb;
  const float fsy = (h > 1) ? (float)(ch - 1) / (float)(h - 1) : 0.f;
  const float fsx = (w > 1) ? (float)(cw - 1) / (float)(w - 1) : 0.f;
  const int sx0 = seg * ADJ_SXT, sx1 = min(sx0 + ADJ_SXT, cw) - 1;
  const int oy_lo = (fsy > 0.f) ? max(0, (int)floorf((float)(sy - 1) / fsy)) : 0;
  const int oy_hi = (fsy > 0.f) ? min(h - 1, (int)ceilf((float)(sy + 1) / fsy)) : h - 1;
  const int ox_lo = (fsx > 0.f) ? max(0, (int)floorf((float)(sx0 - 1) / fsx)) : 0;
  const int ox_hi = (fsx > 0.f) ? min(w - 1, (int)ceilf((float)(sx1 + 1) / fsx)) : w - 1;
  const int n_ox = ox_hi - ox_lo + 1;                       // <= ADJ_MAXOX (checked by the launcher)
  const size_t gbase = ((size_t)b * ncb + cb) * (size_t)h * w * ACT_CB;
  // the rows that contribute (uniform): collected first so that four row loads are in flight per thread
  __shared__ int s_oy[ADJ_MAXR + 4];
  __shared__ float s_wy[ADJ_MAXR + 4];
  __shared__ int s_nr;
  if (tid == 0) {
    int nr = 0;
    for (int oy = oy_lo; oy <= oy_hi && nr < ADJ_MAXR; ++oy) {
      const float fy = fsy * (float)oy;
      const int y0 = min((int)fy, ch - 1), y1 = min(y0 + 1, ch - 1);
      const float ly = fy - (float)y0, hy = 1.f - ly;
      const float wy = (y0 == sy ? hy : 0.f) + (y1 == sy ? ly : 0.f);
      if (wy != 0.f) { s_oy[nr] = oy; s_wy[nr] = wy; ++nr; }
    }
    for (int k = 0; k < 4; ++k) { s_oy[nr + k] = oy_lo; s_wy[nr + k] = 0.f; }      // padding of the last group of four: a valid row, weight 0
    s_nr = nr;
  }
  __syncthreads();
  const int nr = s_nr;
  for (int it = tid; it < n_ox * 4; it += 256) {
    const int oxl = it >> 2, piece = it & 3;
    const uint16_t* col = g + gbase + (size_t)(ox_lo + oxl) * ACT_CB + piece * 8;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int r0 = 0; r0 < nr; r0 += 4) {
      uint4 raw[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) raw[j] = *reinterpret_cast<const uint4*>(col + (size_t)s_oy[r0 + j] * w * ACT_CB);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float f[8];
        unpack8<EK>(raw[j], f);
        const float wy = s_wy[r0 + j];
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = fmaf(wy, f[k], acc[k]);
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) tmp[oxl * 33 + piece * 8 + k] = acc[k];
  }
  __syncthreads();
  for (int it = tid; it < ADJ_SXT * ACT_CB; it += 256) {
    const int sx = sx0 + it % ADJ_SXT, c = it / ADJ_SXT;
    if (sx > sx1) continue;
    const int lo = (fsx > 0.f) ? max(ox_lo, (int)floorf((float)(sx - 1) / fsx)) : ox_lo;
    const int hi = (fsx > 0.f) ? min(ox_hi, (int)ceilf((float)(sx + 1) / fsx)) : ox_hi;
    float acc = 0.f;
    for (int ox = lo; ox <= hi; ++ox) {
      const float fx = fsx * (float)ox;
      const int x0 = min((int)fx, cw - 1), x1 = min(x0 + 1, cw - 1);
      const float lx = fx - (float)x0, hx = 1.f - lx;
      const float wx = (x0 == sx ? hx : 0.f) + (x1 == sx ? lx : 0.f);
      if (wx == 0.f) continue;
      acc = fmaf(wx, tmp[(ox - ox_lo) * 33 + c], acc);
    }
    float* o = dst + (((size_t)b * C + cb * ACT_CB + c) * ch + sy) * cw + sx;
    *o = accumulate ? *o + acc : acc;
  }
}
// generic (any element kind, one thread per output element): the fp32 parity mode
__global__ void upsample_adjoint_view_kernel(ActView g, float* __restrict__ dst, int ch, int cw, int h, int w, int accumulate, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int sx = (int)(i % cw);
  long long r = i / cw;
  const int sy = (int)(r % ch); r /= ch;
  const int c = (int)(r % g.C);
  const long long b = r / g.C;
  const float fsy = (h > 1) ? (float)(ch - 1) / (float)(h - 1) : 0.f;
  const float fsx = (w > 1) ? (float)(cw - 1) / (float)(w - 1) : 0.f;
  const int oy_lo = (fsy > 0.f) ? max(0, (int)floorf((float)(sy - 1) / fsy)) : 0;
  const int oy_hi = (fsy > 0.f) ? min(h - 1, (int)ceilf((float)(sy + 1) / fsy)) : h - 1;
  const int ox_lo = (fsx > 0.f) ? max(0, (int)floorf((float)(sx - 1) / fsx)) : 0;
  const int ox_hi = (fsx > 0.f) ? min(w - 1, (int)ceilf((float)(sx + 1) / fsx)) : w - 1;
  float acc = 0.f;
  for (int oy = oy_lo; oy <= oy_hi; ++oy) {
    const float fy = fsy * (float)oy;
    const int y0 = min((int)fy, ch - 1), y1 = min(y0 + 1, ch - 1);
    const float ly = fy - (float)y0, hy = 1.f - ly;
    const float wy = (y0 == sy ? hy : 0.f) + (y1 == sy ? ly : 0.f);
    if (wy == 0.f) continue;
    for (int ox = ox_lo; ox <= ox_hi; ++ox) {
      const float fx = fsx * (float)ox;
      const int x0 = min((int)fx, cw - 1), x1 = min(x0 + 1, cw - 1);
      const float lx = fx - (float)x0, hx = 1.f - lx;
      const float wx = (x0 == sx ? hx : 0.f) + (x1 == sx ? lx : 0.f);
      if (wx == 0.f) continue;
      acc = fmaf(wy * wx, view_load(g, b, (long long)oy * w + ox, c), acc);
    }
  }
  dst[i] = accumulate ? dst[i] + acc : acc;
}

hipError_t launch_upsample_adjoint(const void* g, int ek, float* dst, int B, int C, int ch, int cw, int h, int w, int accumulate, hipStream_t s, bool tiled) {
  if (C % ACT_CB != 0) return hipErrorInvalidValue;
  if (ek == EK_F32) {
    const long long total = (long long)B * C * ch * cw;
    hipLaunchKernelGGL(upsample_adjoint_view_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       ActView{g, EK_F32, 1, C, (long long)h * w}, dst, ch, cw, h, w, accumulate, total);
    return hipGetLastError();
  }
  {
    // tiled kernel when a segment's destination footprint fits its LDS row (every real configuration: scale factors 2 .. 8)
    const float fsx = (w > 1) ? (float)(cw - 1) / (float)(w - 1) : 0.f;
    const long long span = (fsx > 0.f) ? (long long)std::ceil((double)((cw < ADJ_SXT ? cw : ADJ_SXT) + 1) / (double)fsx) + 3 : (long long)w;
    const int n_seg = (cw + ADJ_SXT - 1) / ADJ_SXT;
    const long long blocks = (long long)B * (C / ACT_CB) * ch * n_seg;
    const float fsy = (h > 1) ? (float)(ch - 1) / (float)(h - 1) : 0.f;
    const long long rows = (fsy > 0.f) ? (long long)std::ceil(2.0 / (double)fsy) + 3 : (long long)h;      // >= the rows with a non-zero weight
    if (tiled && span <= ADJ_MAXOX && rows <= ADJ_MAXR && blocks < (1LL << 31)) {
      if (ek == EK_BF16) hipLaunchKernelGGL(upsample_adjoint_tiled_kernel<EK_BF16>, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const uint16_t*>(g), dst, C, ch, cw, h, w, accumulate, n_seg);
      else hipLaunchKernelGGL(upsample_adjoint_tiled_kernel<EK_F16>, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const uint16_t*>(g), dst, C, ch, cw, h, w, accumulate, n_seg);
      return hipGetLastError();
    }
  }
  const long long total = (long long)B * ch * cw * (C / 8);
  dim3 grid((unsigned)((total + 255) / 256));
  if (ek == EK_BF16) hipLaunchKernelGGL(upsample_adjoint_kernel<EK_BF16>, grid, dim3(256), 0, s, reinterpret_cast<const uint16_t*>(g), dst, C, ch, cw, h, w, accumulate, total);
  else hipLaunchKernelGGL(upsample_adjoint_kernel<EK_F16>, grid, dim3(256), 0, s, reinterpret_cast<const uint16_t*>(g), dst, C, ch, cw, h, w, accumulate, total);
  return hipGetLastError();
}

// dst += src (parameter-gradient sets of the concurrent backward lanes, dd_api.cpp)
__global__ void add_inplace_kernel(float* __restrict__ dst, const float* __restrict__ src, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] += src[i];
}
hipError_t launch_add_inplace(float* dst, const float* src, long long n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(add_inplace_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dst, src, n);
  return hipGetLastError();
}

// dgamma_c += sum_b dgb[b][c][1],  dbeta_c += sum_b dgb[b][c][0]
__global__ void gn_param_grad_kernel(const double* __restrict__ dgb, float* __restrict__ dgamma, float* __restrict__ dbeta, int B, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double sb = 0.0, sg = 0.0;
  for (int b = 0; b < B; ++b) { sb += dgb[((size_t)b * C + c) * 2]; sg += dgb[((size_t)b * C + c) * 2 + 1]; }
  dbeta[c] += (float)sb;
  dgamma[c] += (float)sg;
}
hipError_t launch_gn_param_grad(const double* dgb, float* dgamma, float* dbeta, int B, int C, hipStream_t s) {
  hipLaunchKernelGGL(gn_param_grad_kernel, dim3((C + 63) / 64), dim3(64), 0, s, dgb, dgamma, dbeta, B, C);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Unfused weight gradient (cross-check path):  dW[co][ci][ky][kx] += sum_{b,y,x} g_y[b][y][x][co] * a[b][y+ky-1][x+kx-1][ci].
// One thread per weight element, pixels split over gridDim.y slabs (one fp32 atomic per thread per slab).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) naive_wgrad_kernel(ActView gy, ActView a, float* __restrict__ dw, int h, int w, int B, int rows_per_slab) {
  const int cout = gy.C, cin = a.C;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)cout * cin * 9) return;
  const int kx = (int)(i % 3), ky = (int)((i / 3) % 3);
  const int ci = (int)((i / 9) % cin), co = (int)(i / (9LL * cin));
  const int r0 = blockIdx.y * rows_per_slab, r1 = min(r0 + rows_per_slab, B * h);
  double acc = 0.0;
  for (int r = r0; r < r1; ++r) {
    const int b = r / h, yy = r - b * h;
    const int iy = yy + ky - 1;
    if (iy < 0 || iy >= h) continue;
    float f = 0.f;
    for (int x = 0; x < w; ++x) {
      const int ix = x + kx - 1;
      if (ix < 0 || ix >= w) continue;
      f = fmaf(view_load(gy, b, (long long)yy * w + x, co), view_load(a, b, (long long)iy * w + ix, ci), f);
    }
    acc += f;
  }
  atomicAdd(dw + i, (float)acc);
}
hipError_t launch_naive_wgrad(const ActView& gy, const ActView& a, float* dw_oihw, int B, int h, int w, hipStream_t s) {
  const long long n = (long long)gy.C * a.C * 9;
  const int rows_per_slab = 16;
  dim3 grid((unsigned)((n + 255) / 256), (unsigned)((B * h + rows_per_slab - 1) / rows_per_slab));
  hipLaunchKernelGGL(naive_wgrad_kernel, grid, dim3(256), 0, s, gy, a, dw_oihw, h, w, B, rows_per_slab);
  return hipGetLastError();
}

__global__ void bwd_chain_kernel(float* __restrict__ g, float* __restrict__ ga, const float* __restrict__ c1c2, int k, int mode, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (mode == 0) ga[i] = c1c2[2 * k + 1] * g[i];
  else g[i] = fmaf(c1c2[2 * k], g[i], ga[i]);
}
hipError_t launch_bwd_chain(float* g, float* ga, const float* c1c2, int k, int mode, long long n, hipStream_t s) {
  hipLaunchKernelGGL(bwd_chain_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, g, ga, c1c2, k, mode, n);
  return hipGetLastError();
}

// dst view = src view (layout / element-kind conversion of small tensors: the fp32 state as a conv operand)
__global__ void view_copy_kernel(ActView src, ActView dst, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % src.C);
  const long long bp = i / src.C;
  const long long p = bp % src.HW, b = bp / src.HW;
  view_store(dst, b, p, c, view_load(src, b, p, c));
}
hipError_t launch_view_copy(const ActView& src, const ActView& dst, int B, hipStream_t s) {
  const long long total = (long long)B * src.C * src.HW;
  hipLaunchKernelGGL(view_copy_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, total);
  return hipGetLastError();
}

// dst[b][c][p] (+)= src view  (NCHW fp32 gradient handed back to the caller: g_x, g_cond)
__global__ void view_to_nchw_kernel(ActView v, float* __restrict__ dst, int accumulate, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const long long p = i % v.HW;
  const long long bc = i / v.HW;
  const int c = (int)(bc % v.C);
  const long long b = bc / v.C;
  const float f = view_load(v, b, p, c);
  dst[i] = accumulate ? dst[i] + f : f;
}
hipError_t launch_view_to_nchw(const ActView& v, float* dst, int B, int accumulate, hipStream_t s) {
  const long long total = (long long)B * v.C * v.HW;
  hipLaunchKernelGGL(view_to_nchw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, v, dst, accumulate, total);
  return hipGetLastError();
}

}  // namespace dd
