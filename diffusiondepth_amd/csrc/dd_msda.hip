// dd_msda.hip -- multi-scale deformable attention of the HAHI neck (include/ddepth_msda.h).
//
// Reference: src/model/necks/hahi.py:108-118 builds two mmcv.ops.MultiScaleDeformableAttention modules, :211-223 / :235-247 call them; the
// operator under them (ext_module.ms_deform_attn_forward / _backward of the un-vendored mmcv-full, requirements.txt:84) is Deformable DETR's
// sampling + weighted sum.  It is a gather, not a GEMM: for every (query, head) L x P bilinear samples of a D-channel value row.
//
// Mapping for CDNA: the D channels of one (batch, query, head) are consecutive work-items -- with the neck's D = 512 / 8 = 64 exactly one
// wavefront -- so a sample's location and weight are wave-uniform (one broadcast load each), every corner read is one contiguous 4 D-byte
// row of `value` ([.., head, channel] innermost: coalesced), and the backward's sums over the channels (gradients of a location and of a
// weight are scalars per sample) are wave shuffles, not shared-memory trees.  fp32 throughout, mmcv's evaluation order inside a sample.
#include "../../include/ddepth.h"
#include "../../include/ddepth_msda.h"

#include <hip/hip_runtime.h>

#include <climits>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <string>

namespace {

thread_local std::string g_msda_err;

int msda_fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_msda_err = buf;
  return code;
}

#define MSDA_HIP(expr)                                                                            \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess) return msda_fail(DD_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e));   \
  } while (0)

struct MsdaShape { int B, K, M, D, L, Q, P; };

// the bilinear cell of one sample: corner validity, corner weights, and the four corner rows of value (element offsets of channel 0)
struct Cell {
  bool inside;
  bool v[4];            // (h_low, w_low), (h_low, w_high), (h_high, w_low), (h_high, w_high)
  float w[4];
  float lh, lw, hh, hw;
  long long off[4];
};

__device__ __forceinline__ Cell make_cell(float loc_x, float loc_y, int H, int W, long long level_base, int row_stride) {
  Cell c;
  const float h = loc_y * (float)H - 0.5f, w = loc_x * (float)W - 0.5f;
  c.inside = (h > -1.f) && (w > -1.f) && (h < (float)H) && (w < (float)W);
  const float fh = floorf(h), fw = floorf(w);
  const int hl = (int)fh, wl = (int)fw, hhi = hl + 1, whi = wl + 1;
  c.lh = h - fh; c.lw = w - fw; c.hh = 1.f - c.lh; c.hw = 1.f - c.lw;
  c.v[0] = c.inside && hl >= 0 && wl >= 0;
  c.v[1] = c.inside && hl >= 0 && whi <= W - 1;
  c.v[2] = c.inside && hhi <= H - 1 && wl >= 0;
  c.v[3] = c.inside && hhi <= H - 1 && whi <= W - 1;
  c.w[0] = c.hh * c.hw; c.w[1] = c.hh * c.lw; c.w[2] = c.lh * c.hw; c.w[3] = c.lh * c.lw;
  c.off[0] = level_base + ((long long)hl * W + wl) * row_stride;
  c.off[1] = level_base + ((long long)hl * W + whi) * row_stride;
  c.off[2] = level_base + ((long long)hhi * W + wl) * row_stride;
  c.off[3] = level_base + ((long long)hhi * W + whi) * row_stride;
  return c;
}

__global__ void __launch_bounds__(256) msda_fwd_kernel(MsdaShape s, const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                                       const int64_t* __restrict__ starts, const float* __restrict__ loc,
                                                       const float* __restrict__ attn, float* __restrict__ out, long long total) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % s.D);
  const long long bqm = i / s.D;                    // (b * Q + q) * M + m
  const int m = (int)(bqm % s.M);
  const long long b = bqm / ((long long)s.Q * s.M);
  const int row_stride = s.M * s.D;
  const float* vb = value + (size_t)b * s.K * row_stride + (size_t)m * s.D + c;
  const float* lp = loc + (size_t)bqm * s.L * s.P * 2;
  const float* wp = attn + (size_t)bqm * s.L * s.P;
  float acc = 0.f;
  for (int l = 0; l < s.L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const long long base = (long long)starts[l] * row_stride;
    for (int p = 0; p < s.P; ++p) {
      const float lx = lp[(l * s.P + p) * 2], ly = lp[(l * s.P + p) * 2 + 1], aw = wp[l * s.P + p];
      const Cell cl = make_cell(lx, ly, H, W, base, row_stride);
      if (!cl.inside) continue;
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = cl.v[k] ? vb[cl.off[k]] : 0.f;
      acc += aw * (cl.w[0] * v[0] + cl.w[1] * v[1] + cl.w[2] * v[2] + cl.w[3] * v[3]);
    }
  }
  out[i] = acc;
}

// Four channels per work-item (D % 4 == 0: 16-byte corner loads; with D = 64 a wavefront covers four (query, head) pairs and the cell arithmetic of a
// sample -- identical for all channels -- is amortised over four times the data)
__global__ void __launch_bounds__(256) msda_fwd4_kernel(MsdaShape s, const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                                        const int64_t* __restrict__ starts, const float* __restrict__ loc,
                                                        const float* __restrict__ attn, float* __restrict__ out, long long total4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total4) return;
  const int D4 = s.D >> 2;
  const int c = (int)(i % D4) << 2;
  const long long bqm = i / D4;
  const int m = (int)(bqm % s.M);
  const long long b = bqm / ((long long)s.Q * s.M);
  const int row_stride = s.M * s.D;
  const float* vb = value + (size_t)b * s.K * row_stride + (size_t)m * s.D + c;
  const float* lp = loc + (size_t)bqm * s.L * s.P * 2;
  const float* wp = attn + (size_t)bqm * s.L * s.P;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int l = 0; l < s.L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const long long base = (long long)starts[l] * row_stride;
    for (int p = 0; p < s.P; ++p) {
      const float2 lxy = *reinterpret_cast<const float2*>(lp + (l * s.P + p) * 2);
      const float aw = wp[l * s.P + p];
      const Cell cl = make_cell(lxy.x, lxy.y, H, W, base, row_stride);
      if (!cl.inside) continue;
      float4 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = cl.v[k] ? *reinterpret_cast<const float4*>(vb + cl.off[k]) : make_float4(0.f, 0.f, 0.f, 0.f);
      // (the same order of sums per channel as the one-channel kernel: results are bit-identical)
      acc.x += aw * (cl.w[0] * v[0].x + cl.w[1] * v[1].x + cl.w[2] * v[2].x + cl.w[3] * v[3].x);
      acc.y += aw * (cl.w[0] * v[0].y + cl.w[1] * v[1].y + cl.w[2] * v[2].y + cl.w[3] * v[3].y);
      acc.z += aw * (cl.w[0] * v[0].z + cl.w[1] * v[1].z + cl.w[2] * v[2].z + cl.w[3] * v[3].z);
      acc.w += aw * (cl.w[0] * v[0].w + cl.w[1] * v[1].w + cl.w[2] * v[2].w + cl.w[3] * v[3].w);
    }
  }
  *reinterpret_cast<float4*>(out + (i << 2)) = acc;
}

// SW = the width of the shuffle reduction: the largest power of two <= 64 that divides D (work-items of one (b, q, m) are consecutive, so
// SW-wide groups never straddle a (b, q, m) and never straddle a wavefront)
template <int SW> __device__ __forceinline__ float seg_sum(float v) {
#pragma unroll
  for (int mask = SW / 2; mask >= 1; mask >>= 1) v += __shfl_xor(v, mask);
  return v;
}

template <int SW>
__global__ void __launch_bounds__(256) msda_bwd_kernel(MsdaShape s, const float* __restrict__ value, const int64_t* __restrict__ shapes,
                                                       const int64_t* __restrict__ starts, const float* __restrict__ loc,
                                                       const float* __restrict__ attn, const float* __restrict__ gout,
                                                       float* __restrict__ gvalue, float* __restrict__ gloc, float* __restrict__ gattn,
                                                       long long total) {
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = i0 < total;                   // every work-item stays for the shuffles
  const long long i = live ? i0 : total - 1;
  const int c = (int)(i % s.D);
  const long long bqm = i / s.D;
  const int m = (int)(bqm % s.M);
  const long long b = bqm / ((long long)s.Q * s.M);
  const int row_stride = s.M * s.D;
  const size_t vbase = (size_t)b * s.K * row_stride + (size_t)m * s.D + c;
  const float* lp = loc + (size_t)bqm * s.L * s.P * 2;
  const float* wp = attn + (size_t)bqm * s.L * s.P;
  const float top = live ? gout[i] : 0.f;
  const bool leader = (c % SW) == 0;
  for (int l = 0; l < s.L; ++l) {
    const int H = (int)shapes[2 * l], W = (int)shapes[2 * l + 1];
    const long long base = (long long)starts[l] * row_stride;
    for (int p = 0; p < s.P; ++p) {
      const float lx = lp[(l * s.P + p) * 2], ly = lp[(l * s.P + p) * 2 + 1], aw = wp[l * s.P + p];
      const Cell cl = make_cell(lx, ly, H, W, base, row_stride);
      float val = 0.f, gh = 0.f, gw = 0.f;
      if (cl.inside) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = cl.v[k] ? value[vbase + cl.off[k]] : 0.f;
        val = cl.w[0] * v[0] + cl.w[1] * v[1] + cl.w[2] * v[2] + cl.w[3] * v[3];
        gh = -cl.hw * v[0] - cl.lw * v[1] + cl.hw * v[2] + cl.lw * v[3];
        gw = -cl.hh * v[0] + cl.hh * v[1] - cl.lh * v[2] + cl.lh * v[3];
        if (gvalue && live) {
          const float ta = top * aw;
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (cl.v[k]) atomicAdd(gvalue + vbase + cl.off[k], cl.w[k] * ta);
        }
      }
      // scalars per sample: sums over the channels
      const float ga = seg_sum<SW>(top * val);
      const float gx = seg_sum<SW>((float)W * gw * top * aw);
      const float gy = seg_sum<SW>((float)H * gh * top * aw);
      if (leader && live) {
        const size_t si = (size_t)bqm * s.L * s.P + l * s.P + p;
        if (gattn) atomicAdd(gattn + si, ga);
        if (gloc) { atomicAdd(gloc + 2 * si, gx); atomicAdd(gloc + 2 * si + 1, gy); }
      }
    }
  }
}

int check_msda(MsdaShape& s, int B, int K, int M, int D, int L, int Q, int P, int im2col_step) {
  if (B <= 0 || K <= 0 || M <= 0 || D <= 0 || L <= 0 || Q <= 0 || P <= 0 || im2col_step <= 0)
    return msda_fail(DD_ERR_INVALID_ARG, "sizes and im2col_step must be positive");
  const int step = B < im2col_step ? B : im2col_step;
  if (B % step != 0) return msda_fail(DD_ERR_INVALID_ARG, "batch(%d) must divide im2col_step(%d)", B, step);
  if ((long long)B * Q * M * D > (long long)INT_MAX * 256) return msda_fail(DD_ERR_INVALID_ARG, "output too large");
  s = MsdaShape{B, K, M, D, L, Q, P};
  return DD_OK;
}

}  // namespace

extern "C" {

const char* dd_msda_last_error(void) { return g_msda_err.c_str(); }

int dd_msda_forward(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index, const float* sampling_locations,
                    const float* attention_weights, float* out, int B, int num_keys, int M, int D, int L, int Q, int P, int im2col_step,
                    void* stream) {
  if (!value || !spatial_shapes || !level_start_index || !sampling_locations || !attention_weights || !out)
    return msda_fail(DD_ERR_INVALID_ARG, "null tensor pointer");
  MsdaShape s;
  if (int rc = check_msda(s, B, num_keys, M, D, L, Q, P, im2col_step)) return rc;
  const long long total = (long long)B * Q * M * D;
  const bool vec4 = D % 4 == 0 && ((uintptr_t)value % 16 == 0) && ((uintptr_t)out % 16 == 0) && ((uintptr_t)sampling_locations % 8 == 0);
  if (vec4)
    hipLaunchKernelGGL(msda_fwd4_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, s, value, spatial_shapes,
                       level_start_index, sampling_locations, attention_weights, out, total / 4);
  else
    hipLaunchKernelGGL(msda_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, s, value, spatial_shapes,
                       level_start_index, sampling_locations, attention_weights, out, total);
  MSDA_HIP(hipGetLastError());
  return DD_OK;
}

int dd_msda_backward(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index, const float* sampling_locations,
                     const float* attention_weights, const float* grad_out, float* grad_value, float* grad_sampling_loc,
                     float* grad_attn_weight, int B, int num_keys, int M, int D, int L, int Q, int P, int im2col_step, void* stream) {
  if (!value || !spatial_shapes || !level_start_index || !sampling_locations || !attention_weights || !grad_out)
    return msda_fail(DD_ERR_INVALID_ARG, "null tensor pointer");
  MsdaShape s;
  if (int rc = check_msda(s, B, num_keys, M, D, L, Q, P, im2col_step)) return rc;
  hipStream_t st = (hipStream_t)stream;
  const size_t nsamp = (size_t)B * Q * M * L * P;
  if (grad_value) MSDA_HIP(hipMemsetAsync(grad_value, 0, (size_t)B * num_keys * M * D * sizeof(float), st));
  if (grad_sampling_loc) MSDA_HIP(hipMemsetAsync(grad_sampling_loc, 0, nsamp * 2 * sizeof(float), st));
  if (grad_attn_weight) MSDA_HIP(hipMemsetAsync(grad_attn_weight, 0, nsamp * sizeof(float), st));
  if (!grad_value && !grad_sampling_loc && !grad_attn_weight) return DD_OK;
  const long long total = (long long)B * Q * M * D;
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  int sw = 64;
  while (D % sw != 0) sw >>= 1;
#define MSDA_BWD(SW)                                                                                                                       \
  hipLaunchKernelGGL((msda_bwd_kernel<SW>), grid, block, 0, st, s, value, spatial_shapes, level_start_index, sampling_locations,           \
                     attention_weights, grad_out, grad_value, grad_sampling_loc, grad_attn_weight, total)
  switch (sw) {
    case 64: MSDA_BWD(64); break;
    case 32: MSDA_BWD(32); break;
    case 16: MSDA_BWD(16); break;
    case 8: MSDA_BWD(8); break;
    case 4: MSDA_BWD(4); break;
    case 2: MSDA_BWD(2); break;
    default: MSDA_BWD(1); break;
  }
#undef MSDA_BWD
  MSDA_HIP(hipGetLastError());
  return DD_OK;
}

}  // extern "C"
