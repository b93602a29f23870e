// dd_naive.hip -- DD_PREC_NAIVE_FP32: deliberately simple, unfused fp32 kernels (one thread per
// output element, weights in the reference's own OIHW layout).  They exist to cross-check the fused
// MFMA path ON THE DEVICE layer by layer; they are not a performance path and never a fallback.
#include "dd_kernels.h"

namespace dd {

// out[b][y][x][co] = bias[co] + sum_{ci,ky,kx} in[b][y+ky-1][x+kx-1][ci] * w[co][ci][ky][kx]
__global__ void __launch_bounds__(256) naive_conv3x3_kernel(const float* __restrict__ in, const float* __restrict__ wt,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            int h, int w, int cin, int cout, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int co = (int)(i % cout);
  const long long pix = i / cout;
  const int x = (int)(pix % w);
  const long long t = pix / w;
  const int y = (int)(t % h);
  const long long b = t / h;
  float acc = bias ? bias[co] : 0.f;
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = y + ky - 1;
    if (iy < 0 || iy >= h) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = x + kx - 1;
      if (ix < 0 || ix >= w) continue;
      const float* ip = in + (((size_t)b * h + iy) * w + ix) * cin;
      const float* wp = wt + (size_t)co * cin * 9 + ky * 3 + kx;
      for (int ci = 0; ci < cin; ++ci) acc = fmaf(ip[ci], wp[(size_t)ci * 9], acc);
    }
  }
  out[i] = acc;
}
hipError_t launch_naive_conv3x3(const float* in_nhwc, const float* w_oihw, const float* bias, float* out_nhwc,
                                int B, int h, int w, int cin, int cout, hipStream_t s) {
  const long long total = (long long)B * h * w * cout;
  hipLaunchKernelGGL(naive_conv3x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                     in_nhwc, w_oihw, bias, out_nhwc, h, w, cin, cout, total);
  return hipGetLastError();
}

// GroupNorm statistics of y (NHWC fp32) into slot 0 of stats[b] (fp64 sums).  One block per
// (sample, group); fp64 accumulation throughout.
__global__ void __launch_bounds__(256) naive_gn_stats_kernel(const float* __restrict__ y, double* __restrict__ stats,
                                                             long long HW, int C) {
  __shared__ double sh_s[256], sh_q[256];
  const int b = blockIdx.y, grp = blockIdx.x;
  const int CG = C / GN_GROUPS;
  const long long n = HW * CG;
  double s = 0.0, q = 0.0;
  for (long long i = threadIdx.x; i < n; i += blockDim.x) {
    const long long p = i / CG;
    const int c = grp * CG + (int)(i - p * CG);
    const double v = (double)y[((size_t)b * HW + p) * C + c];
    s += v; q += v * v;
  }
  sh_s[threadIdx.x] = s; sh_q[threadIdx.x] = q;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) { sh_s[threadIdx.x] += sh_s[threadIdx.x + off]; sh_q[threadIdx.x] += sh_q[threadIdx.x + off]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double* dst = stats + (size_t)b * STAT_SLOTS * STAT_STRIDE;   // slot 0; other slots stay zero
    dst[grp * 2] = sh_s[0];
    dst[grp * 2 + 1] = sh_q[0];
  }
}
hipError_t launch_naive_gn_stats(const float* y_nhwc, double* stats, int B, int h, int w, int C, hipStream_t s) {
  hipLaunchKernelGGL(naive_gn_stats_kernel, dim3(GN_GROUPS, (unsigned)B), dim3(256), 0, s, y_nhwc, stats, (long long)h * w, C);
  return hipGetLastError();
}

// out = relu((y - mean) * rstd * gamma + beta) [+ cond + emb[t]]   (literal torch GroupNorm form)
__global__ void __launch_bounds__(256) naive_gn_apply_kernel(const float* __restrict__ y, const double* __restrict__ stats,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ cond, const float* __restrict__ emb,
                                                             const long long* __restrict__ tvec, int t_base, int t_bstride,
                                                             float* __restrict__ out, long long HW, int C, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const long long b = i / ((long long)C * HW);
  const int CG = C / GN_GROUPS;
  const int grp = c / CG;
  const double* st = stats + (size_t)b * STAT_SLOTS * STAT_STRIDE;
  double s = 0.0, q = 0.0;
  for (int k = 0; k < STAT_SLOTS; ++k) { s += st[k * STAT_STRIDE + grp * 2]; q += st[k * STAT_STRIDE + grp * 2 + 1]; }
  const double cnt = (double)HW * CG;
  const double mean = s / cnt;
  double var = q / cnt - mean * mean;
  var = var > 0.0 ? var : 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)GN_EPS));
  float v = (y[i] - (float)mean) * rstd * gamma[c] + beta[c];
  v = (v < 0.f) ? 0.f : v;      // torch.relu: keeps a NaN (fmaxf / v_max_f32 would drop it)
  if (cond) {
    const long long t = clamp_t(tvec[t_base + b * t_bstride]);
    v = (cond[i] + emb[(size_t)t * C + c]) + v;        // feat = feat + E[t]; feat = feat + NE(x)
  }
  out[i] = v;
}
hipError_t launch_naive_gn_apply(const float* y, const double* stats, const float* gamma, const float* beta,
                                 const float* cond, const float* emb, const long long* tvec, int t_base, int t_bstride,
                                 float* out, int B, int h, int w, int C, hipStream_t s) {
  const long long HW = (long long)h * w, total = HW * C * B;
  hipLaunchKernelGGL(naive_gn_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                     y, stats, gamma, beta, cond, emb, tvec, t_base, t_bstride, out, HW, C, total);
  return hipGetLastError();
}

__global__ void naive_axpby_kernel(const float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ c1c2,
                                   int step, float* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = c1c2[2 * step] * x[i] + c1c2[2 * step + 1] * eps[i];
}
hipError_t launch_naive_axpby(const float* x, const float* eps, const float* c1c2, int step, float* out,
                              long long n, hipStream_t s) {
  hipLaunchKernelGGL(naive_axpby_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, x, eps, c1c2, step, out, n);
  return hipGetLastError();
}

}  // namespace dd
