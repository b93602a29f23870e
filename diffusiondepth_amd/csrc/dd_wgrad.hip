// dd_wgrad.hip -- weight gradient of a 3x3 convolution on the matrix cores (what autograd's
// conv2d backward-weight does for the four convs of ScheduledCNNRefine, reference
// src/model/head/ddim_depth_estimate_res.py:303-322, when the reference trains):
//
//     dW[co][ci][ky][kx] += sum_{b, y, x} g_y[b][y][x][co] * a[b][y+ky-1][x+kx-1][ci]
//
// As a GEMM the contraction index is the PIXEL: D[co][ci] += sum_k A[co][k] * B[k][ci] with k = pixel.  The MFMA operand
// registers of a lane hold 8 consecutive k for one row, but both tensors are stored channel-last (a pixel's channels are
// contiguous), so every tile is transposed on its way into LDS:
//
//   gT [64 co][264]          g_y of one 8x32-pixel tile, pixel-contiguous rows (pitch 264: rows shift by 16 B -> conflict-free b128)
//   aT [32 ci][10][40]+pad   the (8+2)x(32+2) input patch, pixel-contiguous rows, channel pitch 520 elements
//
// A workgroup = 6 waves owns (64 output channels) x (32 input channels) x (all 9 taps) and walks a strided set of pixel
// tiles, accumulating in registers; wave (ib, ky) = (32-co half, kernel row) holds the three kx taps (48 accumulators).
// Per 16-pixel k-step a wave reads ONE A fragment and TWO aligned 16-B pieces of the patch row; the kx = 1, 2 operands
// are those two pieces shifted by 2 / 4 bytes in registers (v_alignbyte / renaming), so no unaligned LDS access is needed:
// 3 ds_read_b128 per 3 MFMAs.  At the end every lane adds its 48 values to dW with fp32 atomics (slabs x 18 432 per type).
// bf16 / f16 operands, fp32 accumulation.
#include "dd_elem.h"
#include "dd_gcn.h"

namespace dd {

constexpr int WG_THREADS = 384;
constexpr int GT_PITCH = 264;                    // elements per gT row (256 pixels + 8)
constexpr int AT_ROW = 40;                       // elements per patch row (34 used)
constexpr int AT_PITCH = 520;                    // elements per aT channel (10 x 40 = 400, padded so that rows shift by 16 B)
constexpr int WGRAD_LDS = (64 * GT_PITCH + 32 * AT_PITCH) * 2;

struct WgradParams {
  const void* gy;      // conv-output gradient, activation layout, CO channels
  const void* a;       // conv input activation, activation layout, CI channels
  float* part;         // per-slab partial sums [slab][ky*3+kx][CO][CI] fp32 (plain stores; summed into dW by wgrad_reduce_kernel)
  int CO, CI, B, h, w, tiles_x, tiles_y, slabs;
};

template <int EK>
__global__ void __launch_bounds__(WG_THREADS, 3) wgrad_mfma_kernel(WgradParams p) {
  DD_DYN_SMEM(smem);
  uint16_t* gT = reinterpret_cast<uint16_t*>(smem);
  uint16_t* aT = gT + 64 * GT_PITCH;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ib = wave / 3, ky = wave - ib * 3;
  const int li = lane & 31, g = lane >> 5;
  const int n_cib = (p.CI + 31) / 32;
  const int cog = blockIdx.y / n_cib, cib = blockIdx.y - cog * n_cib;      // 64-co group, 32-ci block
  const int co0 = cog * 64, ci0 = cib * 32;
  const long long HW = (long long)p.h * p.w;
  const int n_tiles = p.B * p.tiles_y * p.tiles_x;

  f32x16_t acc[3];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

  // rows of the LDS images that no channel maps to (CO = 16, CI = 16) stay zero for the whole kernel
  for (int i = tid; i < (64 * GT_PITCH + 32 * AT_PITCH) / 2; i += WG_THREADS) reinterpret_cast<uint32_t*>(smem)[i] = 0u;
  __syncthreads();

  const int co_pieces = min(64, p.CO - co0) / 8;            // 16-B pieces (8 channels) per pixel of the g_y tile: 8 or 2
  const int ci_pieces = min(32, p.CI - ci0) / 8;            // ... of the patch: 4 or 2
  const int co_sh = (co_pieces == 8) ? 3 : 1, ci_sh = (ci_pieces == 4) ? 2 : 1;     // piece counts are 8|2 and 4|2: shifts, not divisions
  const bool ib_live = co0 + ib * 32 < p.CO;                // CO = 16: the upper 32-row half of the A image is all zero
  // staging item = (pair of horizontally adjacent pixels, 16-B piece): two 16-B loads, then per channel ONE 4-byte LDS
  // write holding both pixels (v_perm_b32 interleave) -- half the LDS write instructions of a per-pixel scatter
  constexpr int NG = (128 * 8 + WG_THREADS - 1) / WG_THREADS;   // g_y items per thread (3)
  constexpr int NA = (170 * 4 + WG_THREADS - 1) / WG_THREADS;   // patch items per thread (2): 10 rows x 17 pairs x 4 pieces
  uint4 rg[NG][2], ra[NA][2];

  // global -> registers for one tile (software pipeline: issued before the previous tile's MFMAs, written to LDS after them)
  auto fetch = [&](int tile) {
    const int b = tile / (p.tiles_y * p.tiles_x);
    const int trem = tile - b * p.tiles_y * p.tiles_x;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int y0 = ty * 8, x0 = tx * 32;
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int it = tid + u * WG_THREADS;
      const int q = it & (co_pieces - 1), pair = it >> co_sh;           // pair 0..127: row pair >> 4, columns 2 (pair & 15), +1
      const int gyy = y0 + (pair >> 4), gxx = x0 + 2 * (pair & 15);
      rg[u][0] = make_uint4(0u, 0u, 0u, 0u); rg[u][1] = make_uint4(0u, 0u, 0u, 0u);
      if (it < 128 * co_pieces && gyy < p.h) {
        const int ch = co0 + q * 8;
        const unsigned pstride = (p.CO >= ACT_CB) ? ACT_CB : p.CO;
        const unsigned off = (p.CO >= ACT_CB) ? (((unsigned)(b * (p.CO / ACT_CB) + ch / ACT_CB) * (unsigned)HW + (unsigned)(gyy * p.w + gxx)) * ACT_CB + (ch % ACT_CB))
                                              : (((unsigned)b * (unsigned)HW + (unsigned)(gyy * p.w + gxx)) * p.CO + ch);
        const uint16_t* src = reinterpret_cast<const uint16_t*>(p.gy) + off;
        if (gxx < p.w) rg[u][0] = *reinterpret_cast<const uint4*>(src);
        if (gxx + 1 < p.w) rg[u][1] = *reinterpret_cast<const uint4*>(src + pstride);
      }
    }
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int it = tid + u * WG_THREADS;
      const int q = it & (ci_pieces - 1), pp = it >> ci_sh;             // pp 0..169: patch row pp / 17, column pair pp % 17
      const int pr = pp / 17, pc = 2 * (pp - pr * 17);
      const int gyy = y0 - 1 + pr, gxx = x0 - 1 + pc;
      ra[u][0] = make_uint4(0u, 0u, 0u, 0u); ra[u][1] = make_uint4(0u, 0u, 0u, 0u);
      if (it < 170 * ci_pieces && gyy >= 0 && gyy < p.h) {
        const int ch = ci0 + q * 8;
        const int pstride = (p.CI >= ACT_CB) ? ACT_CB : p.CI;
        const int pix = gyy * p.w + gxx;                                 // may be -1 for the left halo of the first tile column
        const long long off = (p.CI >= ACT_CB) ? (((long long)(b * (p.CI / ACT_CB) + ch / ACT_CB) * HW + pix) * ACT_CB + (ch % ACT_CB))
                                               : (((long long)b * HW + pix) * p.CI + ch);
        const uint16_t* src = reinterpret_cast<const uint16_t*>(p.a) + off;
        if (gxx >= 0 && gxx < p.w) ra[u][0] = *reinterpret_cast<const uint4*>(src);
        if (gxx + 1 >= 0 && gxx + 1 < p.w) ra[u][1] = *reinterpret_cast<const uint4*>(src + pstride);
      }
    }
  };
  // registers -> transposed LDS images: channel k of the pixel pair goes to row k as one dword (pixel 0 low, pixel 1 high)
  auto scatter = [&](const uint4& v0, const uint4& v1, uint32_t* dst, int pitch_dw) {
    const uint32_t a[4] = {v0.x, v0.y, v0.z, v0.w}, c[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      dst[(2 * k) * pitch_dw] = __builtin_amdgcn_perm(c[k], a[k], 0x05040100u);
      dst[(2 * k + 1) * pitch_dw] = __builtin_amdgcn_perm(c[k], a[k], 0x07060302u);
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int u = 0; u < NG; ++u) {
      const int it = tid + u * WG_THREADS;
      if (it < 128 * co_pieces) {
        const int q = it & (co_pieces - 1), pair = it >> co_sh;
        scatter(rg[u][0], rg[u][1], reinterpret_cast<uint32_t*>(gT + (q * 8) * GT_PITCH + 2 * pair), GT_PITCH / 2);
      }
    }
#pragma unroll
    for (int u = 0; u < NA; ++u) {
      const int it = tid + u * WG_THREADS;
      if (it < 170 * ci_pieces) {
        const int q = it & (ci_pieces - 1), pp = it >> ci_sh;
        const int pr = pp / 17, pc = 2 * (pp - pr * 17);
        scatter(ra[u][0], ra[u][1], reinterpret_cast<uint32_t*>(aT + (q * 8) * AT_PITCH + pr * AT_ROW + pc), AT_PITCH / 2);
      }
    }
  };

  // byte offsets from the 16-B aligned LDS base (so that the fragment reads are single ds_read_b128)
  const int arow = ((ib * 32 + li) * GT_PITCH + 8 * g) * 2;
  const int brow = (64 * GT_PITCH + li * AT_PITCH + ky * AT_ROW + 8 * g) * 2;
  int tile = blockIdx.x;
  if (tile < n_tiles) fetch(tile);
  for (; tile < n_tiles; tile += p.slabs) {
    stash();
    __syncthreads();
    if (tile + p.slabs < n_tiles) fetch(tile + p.slabs);      // flies while the MFMAs below run
    // ---- 16 k-steps of 16 pixels: tile row r = s / 2, columns c0 = 16 (s % 2) + 8 g .. + 7 ----
    if (ib_live) {
#pragma unroll 1
      for (int s = 0; s < 16; ++s) {
        const int r = s >> 1, c0 = (s & 1) * 16;
        const uint4 af = *reinterpret_cast<const uint4*>(smem + arow + (r * 32 + c0) * 2);
        const uint4 lo = *reinterpret_cast<const uint4*>(smem + brow + (r * AT_ROW + c0) * 2);
        const uint4 hi = *reinterpret_cast<const uint4*>(smem + brow + (r * AT_ROW + c0 + 8) * 2);
        // patch columns c0+8g+kx .. +7: kx = 0 is `lo`, kx = 1 the 2-byte shift of (lo, hi), kx = 2 the 4-byte shift
        const uint4 b1 = make_uint4(__builtin_amdgcn_alignbyte(lo.y, lo.x, 2), __builtin_amdgcn_alignbyte(lo.z, lo.y, 2),
                                    __builtin_amdgcn_alignbyte(lo.w, lo.z, 2), __builtin_amdgcn_alignbyte(hi.x, lo.w, 2));
        const uint4 b2 = make_uint4(lo.y, lo.z, lo.w, hi.x);
        mma_step<EK>(acc[0], af, lo);
        mma_step<EK>(acc[1], af, b1);
        mma_step<EK>(acc[2], af, b2);
      }
    }
    __syncthreads();                     // everybody is done reading before the next tile overwrites the images
  }
  // ---- flush: D[row co][col ci], lane owns column li and rows 8q + 4g + r.  Plain coalesced stores of this slab's partial
  //      sums (lanes = consecutive ci); fp32 atomics from 512 workgroups onto the same 147k addresses were measured to cost
  //      ~55 us per million, several times the MFMA work itself ----
  const int ci = ci0 + li;
  if (ci < p.CI && ib_live) {
    float* dst = p.part + (size_t)blockIdx.x * 9 * p.CO * p.CI;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int co = co0 + ib * 32 + 8 * q + 4 * g + r;
          if (co < p.CO) dst[((size_t)(ky * 3 + kx) * p.CO + co) * p.CI + ci] = acc[kx][q * 4 + r];
        }
  }
}

// dW[co][ci][tap] += sum over slabs of part[slab][tap][co][ci]
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int slabs, int CO, int CI) {
  const int n = 9 * CO * CI;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;          // index into [tap][co][ci]: consecutive threads = consecutive ci
  if (i >= n) return;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;          // four independent chains: the loads of a chain are issued together
  int sl = 0;
  for (; sl + 4 <= slabs; sl += 4) {
    a0 += part[(size_t)sl * n + i]; a1 += part[(size_t)(sl + 1) * n + i];
    a2 += part[(size_t)(sl + 2) * n + i]; a3 += part[(size_t)(sl + 3) * n + i];
  }
  for (; sl < slabs; ++sl) a0 += part[(size_t)sl * n + i];
  const float acc = (a0 + a1) + (a2 + a3);
  const int ci = i % CI, co = (i / CI) % CO, tap = i / (CI * CO);
  dw[((size_t)co * CI + ci) * 9 + tap] += acc;
}

// second generation (dd_wgrad2.hip): channel counts that are multiples of 64
bool wgrad2_supports(int CO, int CI);
int wgrad2_slabs(int CO, int CI, int n_tiles, int share);
hipError_t launch_wgrad2(const void* gy, const void* a, float* workspace, int ek, int CO, int CI, int B, int h, int w, int* slabs_out, hipStream_t s,
                         int share);

size_t wgrad_workspace_bytes(int CO, int CI, int B, int h, int w) {
  const int types = ((CO + 63) / 64) * ((CI + 31) / 32);
  const int n_tiles = B * ((w + 31) / 32) * ((h + 7) / 8);
  if (wgrad2_supports(CO, CI)) return (size_t)wgrad2_slabs(CO, CI, n_tiles, 1) * 9 * CO * CI * sizeof(float);
  int slabs = (512 + types - 1) / types;
  if (slabs > n_tiles) slabs = n_tiles;
  return (size_t)slabs * 9 * CO * CI * sizeof(float);
}

hipError_t launch_wgrad_mfma(const void* gy, const void* a, float* dw_oihw, float* workspace, int ek, int CO, int CI, int B, int h, int w, hipStream_t s,
                             int share) {
  if (ek != EK_BF16 && ek != EK_F16) return hipErrorInvalidValue;
  if (CO % 8 != 0 || CI % 8 != 0) return hipErrorInvalidValue;
  if ((long long)B * h * w * (CO > CI ? CO : CI) >= (1LL << 32)) return hipErrorInvalidValue;      // 32-bit element offsets inside the kernel
  if (wgrad2_supports(CO, CI)) {
    int slabs2 = 0;
    hipError_t e = launch_wgrad2(gy, a, workspace, ek, CO, CI, B, h, w, &slabs2, s, share);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((9 * CO * CI + 255) / 256), dim3(256), 0, s, workspace, dw_oihw, slabs2, CO, CI);
    return hipGetLastError();
  }
  WgradParams p{};
  p.gy = gy; p.a = a; p.part = workspace; p.CO = CO; p.CI = CI; p.B = B; p.h = h; p.w = w;
  p.tiles_x = (w + 31) / 32; p.tiles_y = (h + 7) / 8;
  const int types = ((CO + 63) / 64) * ((CI + 31) / 32);
  const int n_tiles = B * p.tiles_x * p.tiles_y;
  int slabs = (512 + types - 1) / types;                  // about two workgroups per CU over all types
  if (slabs > n_tiles) slabs = n_tiles;
  if (slabs < 1) slabs = 1;
  p.slabs = slabs;
  static bool attr_set[2] = {false, false};
  const int idx = ek == EK_BF16 ? 0 : 1;
  const void* fn = ek == EK_BF16 ? reinterpret_cast<const void*>(&wgrad_mfma_kernel<EK_BF16>) : reinterpret_cast<const void*>(&wgrad_mfma_kernel<EK_F16>);
  if (!attr_set[idx]) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, WGRAD_LDS);
    if (e != hipSuccess) return e;
    attr_set[idx] = true;
  }
  dim3 grid((unsigned)slabs, (unsigned)types);
  if (ek == EK_BF16) hipLaunchKernelGGL(wgrad_mfma_kernel<EK_BF16>, grid, dim3(WG_THREADS), WGRAD_LDS, s, p);
  else hipLaunchKernelGGL(wgrad_mfma_kernel<EK_F16>, grid, dim3(WG_THREADS), WGRAD_LDS, s, p);
  // rows of `part` that no workgroup writes (ib dead / co >= CO) do not exist: every (tap, co < CO, ci < CI) is written by
  // exactly one wave of one workgroup type per slab
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((9 * CO * CI + 255) / 256), dim3(256), 0, s, workspace, dw_oihw, slabs, CO, CI);
  return hipGetLastError();
}

}  // namespace dd
