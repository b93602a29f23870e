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
  const int slab = 2048;
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
  const long long t = tvec ? tvec[t_base + b * t_bstride] : 0;
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
    const long long row = rows ? rows[t_base + b * t_bstride] : 0;
    atomicAdd(out + (size_t)row * C + tid, (float)t);
  }
}
hipError_t launch_channel_sum(const ActView& v, float* out, const long long* rows, int t_base, int t_bstride, int B, hipStream_t s) {
  if (256 % v.C != 0) return hipErrorInvalidValue;
  const int slab = 2048;
  dim3 grid((unsigned)((v.HW + slab - 1) / slab), (unsigned)B);
  hipLaunchKernelGGL(channel_sum_kernel, grid, dim3(256), 0, s, v, out, rows, t_base, t_bstride, slab);
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
