// dd_wgrad2.hip -- weight gradient of a 3x3 convolution whose channel counts are multiples of 64 (conv2 64->256, conv3 / pred.0 256->64,
// the Swin fuse convs 256->256: 99 % of the backward's weight-gradient FLOPs), second generation:
//
//     dW[co][ci][ky][kx] += sum_{b, y, x} g_y[b][y][x][co] * a[b][y+ky-1][x+kx-1][ci]
//
// The contraction index of this GEMM is the PIXEL, while both tensors are stored channel-last: the MFMA operand registers of a lane want 8
// consecutive pixels of ONE channel.  dd_wgrad.hip (round 1, kept for the 16-channel layers) transposes every tile on its way into LDS
// with VALU permutes and 4-byte LDS writes, 32 workgroup types re-doing it for the same g_y tile: 377 TFLOP/s.  Here nothing is
// transposed in software: gfx950's `ds_read_b64_tr_b16` hands a 16-lane group the transpose of the 16 x 4 elements its lanes point at
// (lane 4 r0 + i receives element i of lanes r0, r0+4, r0+8, r0+12: measured, tools/micro/tr16_probe.hip), so with lane r pointing at
// (pixel p0 + r/4, channels 4 (r%4) .. +3) lane c of the group receives pixels p0 .. p0+3 of channel c -- the operand layout -- straight
// out of an LDS image that is a plain COPY of the channel-blocked global layout ([32-channel block][pixel][32 ch], 64 B per pixel: the
// four pixels a 32-lane half reads are 256 contiguous bytes = every bank once).
//
// A workgroup = 12 waves owns 64 output channels x 64 input channels x all 9 taps and walks a strided set of 8x32-pixel tiles:
// wave (dy, co half, ci half) accumulates dW[32 co][32 ci] for the three taps (dy, 0..2) in 48 registers.  Per 16-pixel k-step a wave
// reads one g_y fragment and three patch fragments (the same pixels shifted by dx: a uniform address offset) = 8 transpose reads for 3
// MFMAs.  The next tile is fetched into registers (7 x 16 B per thread, zero-filled outside the image) while the MFMAs run and written to
// LDS with plain 16-byte stores.  Partial sums per slab go to the workspace of dd_wgrad.hip and its reduce kernel.
#include "dd_elem.h"
#include "dd_gcn.h"

namespace dd {

struct WgradParams {           // as in dd_wgrad.hip
  const void* gy;
  const void* a;
  float* part;
  int CO, CI, B, h, w, tiles_x, tiles_y, slabs;
};

constexpr int W2_THREADS = 768;
constexpr int W2_G_BLOCK = 256 * 64;                        // bytes of one 32-channel block of the g_y tile image (256 pixels x 64 B)
constexpr int W2_P_BLOCK = 340 * 64;                        // ... of the (8+2) x (32+2) input patch
constexpr int W2_LDS = 2 * W2_G_BLOCK + 2 * W2_P_BLOCK;     // 76 288 B
constexpr int W2_PIECES = W2_LDS / 16;                      // 4768 16-byte pieces
constexpr int W2_NPF = (W2_PIECES + W2_THREADS - 1) / W2_THREADS;   // 7 per thread

template <int EK>
__global__ void __launch_bounds__(W2_THREADS, 1) wgrad2_kernel(WgradParams p) {
  DD_DYN_SMEM(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int dy = wave % 3, coh = (wave / 3) & 1, cih = wave / 6;
  const int n_cig = p.CI / 64;
  const int cog = blockIdx.y / n_cig, cig = blockIdx.y - cog * n_cig;
  const int co0 = cog * 64, ci0 = cig * 64;
  const unsigned HW = (unsigned)(p.h * p.w);
  const int tiles_per_img = p.tiles_y * p.tiles_x;
  const int n_tiles = p.B * tiles_per_img;

  f32x16_t acc[3];
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

  // ---- global -> registers: piece `it` of the LDS image (first the two g_y blocks, then the two patch blocks; 4 pieces per pixel) ----
  uint4 pf[W2_NPF];
  auto fetch = [&](int tile) {
    const int b = tile / tiles_per_img;
    const int trem = tile - b * tiles_per_img;
    const int ty = trem / p.tiles_x, tx = trem - ty * p.tiles_x;
    const int y0 = ty * 8, x0 = tx * 32;
#pragma unroll
    for (int u = 0; u < W2_NPF; ++u) {
      const int it = tid + u * W2_THREADS;
      pf[u] = make_uint4(0u, 0u, 0u, 0u);
      if (it < 2048) {                                       // g_y: block it >> 10, pixel (it & 1023) >> 2 of the 8 x 32 tile, piece it & 3
        const int cb = it >> 10, px = (it & 1023) >> 2, q = it & 3;
        const int yy = y0 + (px >> 5), xx = x0 + (px & 31);
        if (yy < p.h && xx < p.w) {
          const unsigned blk = (unsigned)(b * (p.CO >> 5) + (co0 >> 5) + cb);
          pf[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(p.gy) + (blk * HW + (unsigned)(yy * p.w + xx)) * 32u + q * 8);
        }
      } else if (it < W2_PIECES) {                           // patch: block, pixel pr * 34 + pc of the 10 x 34 patch, piece
        const int it2 = it - 2048;
        const int cb = it2 >= 1360 ? 1 : 0, rem = it2 - cb * 1360;
        const int pp = rem >> 2, q = rem & 3;
        const int pr = pp / 34, pc = pp - pr * 34;
        const int yy = y0 - 1 + pr, xx = x0 - 1 + pc;
        if (yy >= 0 && yy < p.h && xx >= 0 && xx < p.w) {
          const unsigned blk = (unsigned)(b * (p.CI >> 5) + (ci0 >> 5) + cb);
          pf[u] = *reinterpret_cast<const uint4*>(reinterpret_cast<const uint16_t*>(p.a) + (blk * HW + (unsigned)(yy * p.w + xx)) * 32u + q * 8);
        }
      }
    }
  };
  auto stash = [&]() {                                       // the image is the piece sequence itself: piece `it` lives at byte 16 it
#pragma unroll
    for (int u = 0; u < W2_NPF; ++u) {
      const int it = tid + u * W2_THREADS;
      if (it < W2_PIECES) *reinterpret_cast<uint4*>(smem + it * 16) = pf[u];
    }
  };

  // per-lane part of the transpose-read addresses: lane (group g = lane >> 4, r = lane & 15) points at pixel + 8 (g >> 1) + (r >> 2),
  // channels 16 (g & 1) + 4 (r & 3) .. +3; the second read of a fragment is 4 pixels (256 B) further
  const int g4 = lane >> 4, r16 = lane & 15;
  const int lane_off = (8 * (g4 >> 1) + (r16 >> 2)) * 64 + (16 * (g4 & 1) + 4 * (r16 & 3)) * 2;
  const int a_base = coh * W2_G_BLOCK + lane_off;
  const int b_base = 2 * W2_G_BLOCK + cih * W2_P_BLOCK + dy * 34 * 64 + lane_off;
  auto frag = [&](int off) {
    const dd_u32x2_t lo = DD_LDS_READ_TR16(smem, off), hi = DD_LDS_READ_TR16(smem, off + 256);
    return make_uint4(lo[0], lo[1], hi[0], hi[1]);
  };

  int tile = blockIdx.x;
  if (tile < n_tiles) fetch(tile);
  for (; tile < n_tiles; tile += p.slabs) {
    stash();
    __syncthreads();
    if (tile + p.slabs < n_tiles) fetch(tile + p.slabs);      // flies while the MFMAs below run
#pragma unroll 2
    for (int s = 0; s < 16; ++s) {                            // k-step: tile row s >> 1, columns 16 (s & 1) .. +15
      const int r = s >> 1, c0 = (s & 1) * 16;
      const uint4 af = frag(a_base + (r * 32 + c0) * 64);
      const int bo = b_base + (r * 34 + c0) * 64;
      const uint4 b0 = frag(bo), b1 = frag(bo + 64), b2 = frag(bo + 128);
      mma_step<EK>(acc[0], af, b0);
      mma_step<EK>(acc[1], af, b1);
      mma_step<EK>(acc[2], af, b2);
    }
    __syncthreads();                     // everybody is done reading before the next tile overwrites the image
  }
  // ---- flush: D[row co][col ci]: lane owns column lane & 31 and rows 8 q + 4 (lane >> 5) + e; coalesced along ci ----
  const int ci = ci0 + cih * 32 + (lane & 31), gq = lane >> 5;
  float* dst = p.part + (size_t)blockIdx.x * 9 * p.CO * p.CI;
#pragma unroll
  for (int dx = 0; dx < 3; ++dx)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int co = co0 + coh * 32 + 8 * q + 4 * gq + e;
        dst[((size_t)(dy * 3 + dx) * p.CO + co) * p.CI + ci] = acc[dx][q * 4 + e];
      }
}

bool wgrad2_supports(int CO, int CI) { return CO % 64 == 0 && CI % 64 == 0; }

int wgrad2_slabs(int CO, int CI, int n_tiles, int share) {
  const int types = (CO / 64) * (CI / 64);
  const int cus = 256 / (share < 1 ? 1 : share);          // `share` concurrent callers (lanes of dd_denoise_backward) split the CUs: fewer slabs
  int slabs = (cus + types - 1) / types;                  // each, i.e. the same partial-sum traffic for the reduction as one caller alone
  if (slabs > n_tiles) slabs = n_tiles;
  return slabs < 1 ? 1 : slabs;
}

hipError_t launch_wgrad2(const void* gy, const void* a, float* workspace, int ek, int CO, int CI, int B, int h, int w, int* slabs_out, hipStream_t s,
                         int share) {
  WgradParams p{};
  p.gy = gy; p.a = a; p.part = workspace; p.CO = CO; p.CI = CI; p.B = B; p.h = h; p.w = w;
  p.tiles_x = (w + 31) / 32; p.tiles_y = (h + 7) / 8;
  p.slabs = wgrad2_slabs(CO, CI, B * p.tiles_x * p.tiles_y, share);
  *slabs_out = p.slabs;
  static bool attr_set[2] = {false, false};
  const int idx = ek == EK_BF16 ? 0 : 1;
  const void* fn = ek == EK_BF16 ? reinterpret_cast<const void*>(&wgrad2_kernel<EK_BF16>) : reinterpret_cast<const void*>(&wgrad2_kernel<EK_F16>);
  if (!attr_set[idx]) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, W2_LDS);
    if (e != hipSuccess) return e;
    attr_set[idx] = true;
  }
  dim3 grid((unsigned)p.slabs, (unsigned)((CO / 64) * (CI / 64)));
  if (ek == EK_BF16) hipLaunchKernelGGL(wgrad2_kernel<EK_BF16>, grid, dim3(W2_THREADS), W2_LDS, s, p);
  else hipLaunchKernelGGL(wgrad2_kernel<EK_F16>, grid, dim3(W2_THREADS), W2_LDS, s, p);
  return hipGetLastError();
}

}  // namespace dd
