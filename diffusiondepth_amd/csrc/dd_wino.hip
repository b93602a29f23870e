// dd_wino.hip -- Winograd F(2x2,3x3) form of a raw-input 256->256 3x3 convolution (the Swin denoiser's upsample_fuse.convB, kernel
// layer 6 of dd_igemm2.hip: no normalisation in front, no GroupNorm behind; reference src/model/head/ddim_depth_estimate_res_swin_addHAHI.py:
// 328-333), 16-bit operand modes only.  EXPERIMENTAL, OFF BY DEFAULT (dd_set_option("winograd", 1)): written in round 1 without GPU time left
// to run it -- DESIGN.md section 7 item 0 has the motivation (the direct kernels sit at the socket power cap; 16 instead of 36 multiplies per
// 2x2 output tile), the measured instruction-mix ceiling (tools/micro/mfma_power.hip mode 3) and the numerics (tools/winograd_numerics.py).
// First GPU contact (run 40): parity as predicted (eps error 1.09x the direct kernel's), but 2.3x SLOWER than the direct convB: still written
// for being right, not fast -- single-buffered LDS, three barriers per 16-channel chunk, the input transform repeated by the 4 cout splits.
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A        d = 4x4 input tile (stride 2), Y = 2x2 output tile
//
// Workgroup: 512 threads = 8 waves, 8x32 output pixels = 4x16 tiles = 64 tiles, 64 output channels (4 cout splits).  Per 16-channel chunk
//   (a) the 10x34-pixel raw patch (16 ch) and the chunk's pre-transformed weights U[16 pos][64 co][16 ch] go to LDS,
//   (b) lane (tile, channel pair) forms V = B^T d B in fp32 and writes V[16 pos][64 tiles][16 ch] (rounded to the operand type),
//   (c) wave q multiplies positions 2q and 2q+1:  M_pos[co][tile] += U_pos[co][ch] . V_pos[ch][tile]  -- 2x2 blocks of v_mfma_f32_32x32x16,
//       weights as the A operand, tiles as the B operand (dd_elem.h mma_step), 128 accumulator registers per lane.
// Epilogue: the 16 position accumulators meet in LDS (fp32, one 32 co x 32 tile block at a time) and 128 lanes apply A^T M A, add the bias
// and store 4 pixels x 8 couts (two 8-byte stores per pixel into the channel-blocked activation layout).
#include "dd_elem.h"
#include "dd_gcn.h"     // LDS-DMA / s_waitcnt / LDS base address; tests/host_emul compiles this file for the host through the same macros

namespace dd {

namespace {

constexpr int W_TH = 8, W_TW = 32;                 // output pixels per workgroup
constexpr int W_TY = W_TH / 2, W_TX = W_TW / 2;    // 4 x 16 Winograd tiles
constexpr int W_TILES = W_TY * W_TX;               // 64
constexpr int W_PH = W_TH + 2, W_PW = W_TW + 2;    // 10 x 34 input patch
constexpr int W_CK = 16;                           // channels per chunk = one MFMA k-step
constexpr int W_NT = 64;                           // output channels per workgroup
constexpr int W_THREADS = 512;
constexpr int W_RAW_BYTES = W_PH * W_PW * W_CK * 2;        // 10880
constexpr int W_V_BYTES = 16 * W_TILES * W_CK * 2;         // 32768
constexpr int W_U_BYTES = 16 * W_NT * W_CK * 2;            // 32768
constexpr int W_V_OFF = 10944;                             // RAW region padded to a multiple of 64 B
constexpr int W_U_OFF = W_V_OFF + W_V_BYTES;
constexpr int W_SMEM = W_U_OFF + W_U_BYTES;                // 76480 B; the epilogue's 64 KB fp32 block reuses [0, 65536)
static_assert(W_RAW_BYTES <= W_V_OFF && 16 * 32 * 32 * 4 <= W_SMEM, "LDS carve");

template <int EK> __device__ __forceinline__ float ld16(uint32_t half) {
  return (EK == EK_BF16) ? bf16_to_f32(half & 0xFFFFu) : f16_to_f32(half & 0xFFFFu);
}

}  // namespace

template <int EK>
__global__ void __launch_bounds__(W_THREADS) conv_wino_raw_kernel(ConvParams p) {
  static_assert(EK == EK_BF16 || EK == EK_F16, "16-bit operand modes only");
  constexpr int CIN = COND_C, COUT = COND_C, NCHUNK = CIN / W_CK, NSPLIT = COUT / W_NT;
  DD_DYN_SMEM(smem);
  char* s_raw = smem;
  char* s_v = smem + W_V_OFF;
  char* s_u = smem + W_U_OFF;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, g = lane >> 5;
  const int h = p.h, w = p.w;
  const int wgid = blockIdx.x;
  const int nsplit = wgid % NSPLIT;
  const int tile_id = wgid / NSPLIT;
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const int b = tile_id / tiles_per_img;
  const int trem = tile_id - b * tiles_per_img;
  const int y0 = (trem / p.tiles_x) * W_TH, x0 = (trem % p.tiles_x) * W_TW;
  const size_t HW = (size_t)h * w;
  const char* in_b = reinterpret_cast<const char*>(p.in) + (size_t)b * HW * CIN * 2;
  const char* u_g = reinterpret_cast<const char*>(p.wpack) + (size_t)nsplit * NCHUNK * W_U_BYTES;

  f32x16_t acc[2][2][2];          // [position of this wave][cout block][tile block]
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][m][n][r] = 0.f;

  // transform role of this lane: tile tt, channels 2*cp and 2*cp+1 of the chunk
  const int cp = tid & 7, tt = tid >> 3;
  const int tty = tt / W_TX, ttx = tt - tty * W_TX;

  // raw patch and weights of a chunk travel global -> registers -> LDS, one chunk ahead: the loads of chunk c+1 are in flight while
  // chunk c is transformed and multiplied (without this every chunk exposed a full global-load latency: 1214 us per KITTI B=4 launch)
  static_assert(W_U_BYTES / 16 / W_THREADS == 4, "four weight pieces per lane");
  uint4 r0, r1, u0, u1, u2, u3;      // named registers (an indexed array behind lambdas ended up in scratch memory)
  // per-lane patch items (pixel, 16-byte half) do not depend on the chunk
  const int item1 = tid + W_THREADS;
  const bool have1 = item1 < W_PH * W_PW * 2;
  auto item_geom = [&](int item, size_t& goff, bool& inside) {
    const int pp = item >> 1, hf = item & 1;
    const int pr = pp / W_PW, pc = pp - pr * W_PW;
    const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
    inside = gy >= 0 && gy < h && gx >= 0 && gx < w;
    const int gyc = min(max(gy, 0), h - 1), gxc = min(max(gx, 0), w - 1);            // clamped: unconditional load, select afterwards
    goff = ((size_t)gyc * w + gxc) * ACT_CB * 2 + hf * 16;
  };
  size_t goff0, goff1;
  bool in0, in1;
  item_geom(tid, goff0, in0);
  item_geom(have1 ? item1 : 0, goff1, in1);
  const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
  auto gload = [&](int chunk) {
    // activation layout [B][C/32][h][w][32]: the chunk's 16 channels are 32 contiguous bytes per pixel
    const char* cb = in_b + (((size_t)(chunk >> 1) * HW) * ACT_CB + (size_t)(chunk & 1) * W_CK) * 2;
    const uint4 a0 = *reinterpret_cast<const uint4*>(cb + goff0);
    const uint4 a1 = *reinterpret_cast<const uint4*>(cb + goff1);
    r0 = a0;                                                   // the zero-padding select waits for lstore(): selecting here would make
    r1 = a1;                                                   // the wave wait for the loads at once and the prefetch would be none
    const char* src = u_g + (size_t)chunk * W_U_BYTES + (size_t)tid * 16;
    u0 = *reinterpret_cast<const uint4*>(src);
    u1 = *reinterpret_cast<const uint4*>(src + (size_t)W_THREADS * 16);
    u2 = *reinterpret_cast<const uint4*>(src + (size_t)2 * W_THREADS * 16);
    u3 = *reinterpret_cast<const uint4*>(src + (size_t)3 * W_THREADS * 16);
  };
  auto lstore = [&]() {
    *reinterpret_cast<uint4*>(s_raw + (tid >> 1) * (W_CK * 2) + (tid & 1) * 16) = in0 ? r0 : zero4;      // zero padding of the convolution
    if (have1) *reinterpret_cast<uint4*>(s_raw + (item1 >> 1) * (W_CK * 2) + (item1 & 1) * 16) = in1 ? r1 : zero4;
    *reinterpret_cast<uint4*>(s_u + tid * 16) = u0;
    *reinterpret_cast<uint4*>(s_u + (tid + W_THREADS) * 16) = u1;
    *reinterpret_cast<uint4*>(s_u + (tid + 2 * W_THREADS) * 16) = u2;
    *reinterpret_cast<uint4*>(s_u + (tid + 3 * W_THREADS) * 16) = u3;
  };
  gload(0);
#pragma unroll 1
  for (int chunk = 0; chunk < NCHUNK; ++chunk) {
    // ---- (a) this chunk's raw patch + weights: registers -> LDS; next chunk's loads start ----------------------------------
    lstore();
    __syncthreads();
    if (chunk + 1 < NCHUNK) gload(chunk + 1);
    // ---- (b) input transform V = B^T d B of tile tt, channel pair cp (fp32, rounded once on the way out) ----------------
    {
      float d[2][4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint32_t pk = *reinterpret_cast<const uint32_t*>(s_raw + ((2 * tty + i) * W_PW + 2 * ttx + j) * (W_CK * 2) + cp * 4);
          d[0][i][j] = ld16<EK>(pk);
          d[1][i][j] = ld16<EK>(pk >> 16);
        }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float t[4][4];                                          // t = B^T d : rows (d0-d2, d1+d2, d2-d1, d1-d3)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          t[0][j] = d[c][0][j] - d[c][2][j];
          t[1][j] = d[c][1][j] + d[c][2][j];
          t[2][j] = d[c][2][j] - d[c][1][j];
          t[3][j] = d[c][1][j] - d[c][3][j];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {                           // V = t B : columns likewise
          d[c][i][0] = t[i][0] - t[i][2];
          d[c][i][1] = t[i][1] + t[i][2];
          d[c][i][2] = t[i][2] - t[i][1];
          d[c][i][3] = t[i][1] - t[i][3];
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<uint32_t*>(s_v + (((i * 4 + j) * W_TILES + tt) * W_CK + cp * 2) * 2) = pack2<EK>(d[0][i][j], d[1][i][j]);
    }
    __syncthreads();
    // ---- (c) positions 2*wave, 2*wave+1:  M[co][tile] += U[co][ch] . V[ch][tile] ----------------------------------------
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int pos = 2 * wave + a;
      uint4 wf[2], vf[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) wf[m] = *reinterpret_cast<const uint4*>(s_u + ((pos * W_NT + m * 32 + li) * W_CK + g * 8) * 2);
#pragma unroll
      for (int n = 0; n < 2; ++n) vf[n] = *reinterpret_cast<const uint4*>(s_v + ((pos * W_TILES + n * 32 + li) * W_CK + g * 8) * 2);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) mma_step<EK>(acc[a][m][n], wf[m], vf[n]);
    }
    __syncthreads();                                            // everybody is done with RAW / U / V of this chunk
  }

  // ---- epilogue: Y = A^T M A per (cout, tile), one 32 co x 32 tile block at a time through LDS ----------------------------
  float* s_m = reinterpret_cast<float*>(smem);                  // [16 pos][32 tiles][32 co] fp32 = 64 KB
  char* out_b = reinterpret_cast<char*>(p.out) + (size_t)b * HW * COUT * 2;
#pragma unroll 1
  for (int blk = 0; blk < 4; ++blk) {
    const int m = blk >> 1, n = blk & 1;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int pos = 2 * wave + a;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        // accumulator register r = 4q+i of lane (li, g): cout 8q + 4g + i of the block, tile li (dd_igemm2.hip epilogue convention)
        float4 v;
        if (m == 0 && n == 0) v = make_float4(acc[a][0][0][q * 4], acc[a][0][0][q * 4 + 1], acc[a][0][0][q * 4 + 2], acc[a][0][0][q * 4 + 3]);
        else if (m == 0) v = make_float4(acc[a][0][1][q * 4], acc[a][0][1][q * 4 + 1], acc[a][0][1][q * 4 + 2], acc[a][0][1][q * 4 + 3]);
        else if (n == 0) v = make_float4(acc[a][1][0][q * 4], acc[a][1][0][q * 4 + 1], acc[a][1][0][q * 4 + 2], acc[a][1][0][q * 4 + 3]);
        else v = make_float4(acc[a][1][1][q * 4], acc[a][1][1][q * 4 + 1], acc[a][1][1][q * 4 + 2], acc[a][1][1][q * 4 + 3]);
        *reinterpret_cast<float4*>(s_m + ((pos * 32 + li) * 32 + 8 * q + 4 * g)) = v;
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int tj = tid >> 2, cg = tid & 3;                    // tile tj of the block, couts 8*cg .. 8*cg+7
      const int T = n * 32 + tj;
      const int ty = T / W_TX, tx = T - ty * W_TX;
      const int co = nsplit * W_NT + m * 32 + cg * 8;
#pragma unroll
      for (int c8 = 0; c8 < 2; ++c8) {                          // four couts at a time keeps the transform at ~50 live registers
        float t0[4][4], t1[4][4];                               // A^T M, accumulated row by row: rows (m0+m1+m2, m1-m2-m3); [col][cout]
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float4 v4 = *reinterpret_cast<const float4*>(s_m + (((i * 4 + j) * 32 + tj) * 32 + cg * 8 + c8 * 4));
            const float mv[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              if (i == 0) { t0[j][c] = mv[c]; }
              else if (i == 1) { t0[j][c] += mv[c]; t1[j][c] = mv[c]; }
              else if (i == 2) { t0[j][c] += mv[c]; t1[j][c] -= mv[c]; }
              else { t1[j][c] -= mv[c]; }
            }
          }
        const float4 b4 = *reinterpret_cast<const float4*>(p.bias + co + c8 * 4);
        const float bias[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
          for (int dx = 0; dx < 2; ++dx) {
            const int gy = y0 + 2 * ty + dy, gx = x0 + 2 * tx + dx;
            if (gy < h && gx < w) {
              float v[4];
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const float* tr = dy == 0 ? &t0[0][0] : &t1[0][0];
                // (A^T M) A : columns (c0+c1+c2, c1-c2-c3)
                v[c] = (dx == 0 ? (tr[0 * 4 + c] + tr[1 * 4 + c] + tr[2 * 4 + c]) : (tr[1 * 4 + c] - tr[2 * 4 + c] - tr[3 * 4 + c])) + bias[c];
              }
              *reinterpret_cast<uint2*>(out_b + act_offset(COUT, h, w, 0, co + c8 * 4, gy, gx) * 2) = make_uint2(pack2<EK>(v[0], v[1]), pack2<EK>(v[2], v[3]));
            }
          }
      }
    }
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------------------------------
// v2 (option "winograd" = 2; NEVER RUN: written after the last GPU second of round 1): the same arithmetic and the same LDS images, but
// double-buffered so that the two heavy phases overlap inside every wave: while the matrix pipe works on chunk c (fragments of U[c], V[c]
// read, 8 MFMAs issued), the VALU transforms chunk c+1 (raw[c+1] -> V[c+1]); two barriers per chunk instead of three, none between the
// MFMAs and the transform.  v1 measured 1131 us per KITTI B=4 launch against 494 us for the direct kernel: its three barrier-separated
// phases (LDS fill, VALU transform, MFMA) never overlap and only one workgroup fits a CU.
//   buffers of chunk k: raw[k & 1], U[k & 1], V[k & 1]
//   iteration c:  registers(chunk c+1) -> raw / U [(c+1) & 1];  barrier A;  global loads of chunk c+2 -> registers;
//                 MFMAs of chunk c;  transform raw[(c+1) & 1] -> V[(c+1) & 1];  barrier B
//   (raw / U [(c+1)&1] were last read in iteration c-2 / c-1 before their barrier B; V[(c+1)&1] was last read by the MFMAs of c-1)
// ------------------------------------------------------------------------------------------------------------------------------------
namespace {
constexpr int W2_RAW_STRIDE = W_V_OFF;                          // 10944
constexpr int W2_V_OFF = 2 * W2_RAW_STRIDE;                     // 21888
constexpr int W2_U_OFF = W2_V_OFF + 2 * W_V_BYTES;              // 87424
constexpr int W2_SMEM = W2_U_OFF + 2 * W_U_BYTES;               // 152960 B of the CU's 160 KB
static_assert(W2_SMEM <= 160 * 1024 && 16 * 32 * 32 * 4 <= W2_SMEM, "LDS carve (v2)");
}  // namespace

// PK (f16 only, option "winograd" = 3): the input transform's 64 additions run as 32 packed f16 adds on the channel PAIR (v_pk_add_f16),
// without unpacking to fp32 and re-packing: ~60 instead of ~150 VALU instructions per lane and chunk.  Every add rounds to f16;
// tools/winograd_numerics.py puts the cost at +4.5 % depth RMSE (2.63e-4 instead of 2.52e-4; direct f16 2.11e-4).
// Generalised (option "winograd" = 4 / 5, NEVER RUN) to every large convolution of both denoisers:
//   CIN, COUT  64->256 (conv2), 256->64 (conv3, Swin pred.0), 256->256 (Swin convA / convB)
//   PRO        PRO_RAW: operands as stored | PRO_GN: relu(a*y + b) | PRO_GN_ADD: relu(a*y + b) + cond + e   -- applied when the prefetched
//              registers go to LDS, from the per-(image, channel) table (a, b, e) that wino_gn_table_kernel derives from the producer's
//              GroupNorm partial sums (p.wino_tab: [B][CIN][4] floats); zero padding applies AFTER the normalisation
//   STATS      Sigma / Sigma x^2 of this layer's fp32 outputs per GroupNorm group -> one fp64 atomic per group per workgroup (slot layout of
//              dd_kernels.h, as dd_igemm2.hip's epilogue)
template <int EK, bool PK, int CIN, int COUT, int PRO, bool STATS>
__global__ void __launch_bounds__(W_THREADS) conv_wino_raw_v2_kernel(ConvParams p) {
  static_assert(EK == EK_BF16 || EK == EK_F16, "16-bit operand modes only");
  static_assert(!PK || EK == EK_F16, "packed transform adds exist for f16 only");
  static_assert(CIN % 32 == 0 && COUT % W_NT == 0 && (COUT == 64 || COUT == 256), "channel-blocked layouts, 64 couts per workgroup");
  constexpr int NCHUNK = CIN / W_CK, NSPLIT = COUT / W_NT;
  DD_DYN_SMEM(smem);
  float* s_tab = reinterpret_cast<float*>(smem + W2_SMEM);        // [3][CIN]: a, b, e of the prologue (PRO != PRO_RAW)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, g = lane >> 5;
  const int h = p.h, w = p.w;
  const int wgid = blockIdx.x;
  // p.wino_flags bit 0 (option "winograd_dma"): the weight images go global -> LDS by LDS-DMA instead of through
  // registers and six ds_write_b128 per lane (13 LDS-path cycles each, MI355X_MICROARCH.md section LDS)
  const bool use_dma = (p.wino_flags & 1) != 0;
  const unsigned lds_base = DD_LDS_BASE(smem);
  const int nsplit = wgid % NSPLIT;
  const int tile_id = wgid / NSPLIT;
  const int tiles_per_img = p.tiles_x * p.tiles_y;
  const int b = tile_id / tiles_per_img;
  const int trem = tile_id - b * tiles_per_img;
  const int y0 = (trem / p.tiles_x) * W_TH, x0 = (trem % p.tiles_x) * W_TW;
  const size_t HW = (size_t)h * w;
  const char* in_b = reinterpret_cast<const char*>(p.in) + (size_t)b * HW * CIN * 2;
  const char* u_g = reinterpret_cast<const char*>(p.wpack) + (size_t)nsplit * NCHUNK * W_U_BYTES;
  const char* cond_b = (PRO == PRO_GN_ADD) ? reinterpret_cast<const char*>(p.cond) + (size_t)b * HW * CIN * 2 : nullptr;
  if constexpr (PRO != PRO_RAW) {
    if (tid < CIN) {
      const float4 t4 = reinterpret_cast<const float4*>(p.wino_tab)[(size_t)b * CIN + tid];
      s_tab[tid] = t4.x; s_tab[CIN + tid] = t4.y; s_tab[2 * CIN + tid] = t4.z;
    }                                                           // visible after the first barrier below
  }

  f32x16_t acc[2][2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][m][n][r] = 0.f;

  const int cp = tid & 7, tt = tid >> 3;
  const int tty = tt / W_TX, ttx = tt - tty * W_TX;

  uint4 r0, r1, u0, u1, u2, u3;
  uint4 c0 = make_uint4(0u, 0u, 0u, 0u), c1 = c0;                // PRO_GN_ADD: the condition map's pieces
  int loaded_chunk = 0;                                         // chunk whose pieces the registers hold (prologue table index)
  const int item1 = tid + W_THREADS;
  const bool have1 = item1 < W_PH * W_PW * 2;
  auto item_geom = [&](int item, size_t& goff, bool& inside) {
    const int pp = item >> 1, hf = item & 1;
    const int pr = pp / W_PW, pc = pp - pr * W_PW;
    const int gy = y0 - 1 + pr, gx = x0 - 1 + pc;
    inside = gy >= 0 && gy < h && gx >= 0 && gx < w;
    const int gyc = min(max(gy, 0), h - 1), gxc = min(max(gx, 0), w - 1);
    goff = ((size_t)gyc * w + gxc) * ACT_CB * 2 + hf * 16;
  };
  size_t goff0, goff1;
  bool in0, in1;
  item_geom(tid, goff0, in0);
  item_geom(have1 ? item1 : 0, goff1, in1);
  auto gload = [&](int chunk) {
    const char* cb = in_b + (((size_t)(chunk >> 1) * HW) * ACT_CB + (size_t)(chunk & 1) * W_CK) * 2;
    const uint4 a0 = *reinterpret_cast<const uint4*>(cb + goff0);
    const uint4 a1 = *reinterpret_cast<const uint4*>(cb + goff1);
    r0 = a0;                                                    // selects deferred to lstore() (a select here stalls on the load at once)
    r1 = a1;
    if constexpr (PRO == PRO_GN_ADD) {
      const char* cc = cond_b + (((size_t)(chunk >> 1) * HW) * ACT_CB + (size_t)(chunk & 1) * W_CK) * 2;
      c0 = *reinterpret_cast<const uint4*>(cc + goff0);
      c1 = *reinterpret_cast<const uint4*>(cc + goff1);
    }
    loaded_chunk = chunk;
    if (!use_dma) {
      const char* src = u_g + (size_t)chunk * W_U_BYTES + (size_t)tid * 16;
      u0 = *reinterpret_cast<const uint4*>(src);
      u1 = *reinterpret_cast<const uint4*>(src + (size_t)W_THREADS * 16);
      u2 = *reinterpret_cast<const uint4*>(src + (size_t)2 * W_THREADS * 16);
      u3 = *reinterpret_cast<const uint4*>(src + (size_t)3 * W_THREADS * 16);
    }
  };
  // LDS-DMA of one chunk's weight image (32 KB = 32 KiB-pieces; wave q copies pieces q, q+8, q+16, q+24; lane l lands at M0 + 16 l).
  // Issued through inline asm exactly as dd_igemm2.hip's weight ring (hipcc would drain vmcnt(0) behind the builtin); hipcc does not count
  // these VMEM operations, so the wait for them is the explicit vmcnt(0) in front of the barrier that publishes them.  The k-half swizzle
  // is applied on the SOURCE side: LDS piece P receives packed-image piece P ^ ((P >> 4) & 1).
  auto dma_u = [&](int chunk, int buf) {
    const char* src = u_g + (size_t)chunk * W_U_BYTES;
#pragma unroll
    for (int c = 0; c < W_U_BYTES / 1024 / (W_THREADS / 64); ++c) {
      const int kc = c * (W_THREADS / 64) + wave;
      const int P = kc * 64 + lane;
      const char* gsrc = src + (size_t)(P ^ ((P >> 4) & 1)) * 16;
      const unsigned ldst = __builtin_amdgcn_readfirstlane(lds_base + W2_U_OFF + buf * W_U_BYTES + kc * 1024);
      DD_LDS_DMA16(smem, gsrc, ldst);
    }
  };
  // prologue of one 16-byte piece (8 channels starting at ch0 of one pixel): relu(a*y + b) [+ cond + e]; outside the image: zero
  auto prologue = [&](const uint4& raw, const uint4& cnd, bool inside, int ch0) -> uint4 {
    if constexpr (PRO == PRO_RAW) { (void)cnd; (void)ch0; return inside ? raw : make_uint4(0u, 0u, 0u, 0u); }   // zero padding of the convolution
    float v[8], ta[8], tb[8];
    Piece<EK>::unpack(raw, v);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const float4 a4 = *reinterpret_cast<const float4*>(s_tab + ch0 + 4 * q);
      const float4 b4 = *reinterpret_cast<const float4*>(s_tab + CIN + ch0 + 4 * q);
      ta[4 * q] = a4.x; ta[4 * q + 1] = a4.y; ta[4 * q + 2] = a4.z; ta[4 * q + 3] = a4.w;
      tb[4 * q] = b4.x; tb[4 * q + 1] = b4.y; tb[4 * q + 2] = b4.z; tb[4 * q + 3] = b4.w;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = fmaxf(fmaf(ta[i], v[i], tb[i]), 0.f);
    if constexpr (PRO == PRO_GN_ADD) {
      float cv[8], te[8];
      Piece<EK>::unpack(cnd, cv);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const float4 e4 = *reinterpret_cast<const float4*>(s_tab + 2 * CIN + ch0 + 4 * q);
        te[4 * q] = e4.x; te[4 * q + 1] = e4.y; te[4 * q + 2] = e4.z; te[4 * q + 3] = e4.w;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = v[i] + (cv[i] + te[i]);        // same association as dd_igemm2.hip's transform_item
    }
    const uint4 o = Piece<EK>::pack(v);
    return inside ? o : make_uint4(0u, 0u, 0u, 0u);
  };
  // raw image in LDS: pixel (pr, pc) lives in slot pr * 34 + (pc ^ ((pc >> 2) & 1)).  With linear columns the transform's ds_read_b32 (banks
  // = dword mod 32, lane groups of 32 = 4 tiles x 8 channel pairs, tiles 2 pixels = 16 dwords apart) hit every bank twice; swapping the
  // column pairs (4,5) (6,7) of every 8 makes the four tiles' pixels differ mod 4 for every j (checked exhaustively: 2-way -> 1-way; the
  // 16-byte stores below stay conflict-free)
  auto raw_slot = [&](int item) {
    const int pp = item >> 1;
    const int pr = pp / W_PW, pc = pp - pr * W_PW;
    return (pr * W_PW + (pc ^ ((pc >> 2) & 1))) * (W_CK * 2) + (item & 1) * 16;
  };
  const int slot0 = raw_slot(tid), slot1 = raw_slot(have1 ? item1 : 0);
  auto lstore = [&](int buf) {
    char* s_raw = smem + buf * W2_RAW_STRIDE;
    char* s_u = smem + W2_U_OFF + buf * W_U_BYTES;
    const int chb = loaded_chunk * W_CK;
    *reinterpret_cast<uint4*>(s_raw + slot0) = prologue(r0, c0, in0, chb + (tid & 1) * 8);
    if (have1) *reinterpret_cast<uint4*>(s_raw + slot1) = prologue(r1, c1, in1, chb + (item1 & 1) * 8);
    // U / V images: row r (cout or tile, 0..63 inside a position) holds its two 16-byte k-halves swapped when bit 3 of r is set: the
    // fragment ds_read_b128 (banks = dword mod 64, lane groups {0-3,12-15,20-27} {4-11,16-19,28-31} ...) then touch every bank once per group
    // (linear rows: twice; checked with the bank calculator).  Piece q of the packed image = (row = q >> 1, half = q & 1).
    if (!use_dma) {
      *reinterpret_cast<uint4*>(s_u + ((tid) ^ ((tid >> 4) & 1)) * 16) = u0;
      *reinterpret_cast<uint4*>(s_u + ((tid + W_THREADS) ^ ((tid >> 4) & 1)) * 16) = u1;
      *reinterpret_cast<uint4*>(s_u + ((tid + 2 * W_THREADS) ^ ((tid >> 4) & 1)) * 16) = u2;
      *reinterpret_cast<uint4*>(s_u + ((tid + 3 * W_THREADS) ^ ((tid >> 4) & 1)) * 16) = u3;
    }
  };
  auto transform = [&](int buf) {                               // raw[buf] -> V[buf]: tile tt, channel pair cp
    const char* s_raw = smem + buf * W2_RAW_STRIDE;
    char* s_v = smem + W2_V_OFF + buf * W_V_BYTES;
    if constexpr (PK) {
      f16x2_t e[4][4], t[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          e[i][j] = __builtin_bit_cast(f16x2_t, *reinterpret_cast<const uint32_t*>(s_raw + ((2 * tty + i) * W_PW + ((2 * ttx + j) ^ (((2 * ttx + j) >> 2) & 1))) * (W_CK * 2) + cp * 4));
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t[0][j] = e[0][j] - e[2][j];
        t[1][j] = e[1][j] + e[2][j];
        t[2][j] = e[2][j] - e[1][j];
        t[3][j] = e[1][j] - e[3][j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        e[i][0] = t[i][0] - t[i][2];
        e[i][1] = t[i][1] + t[i][2];
        e[i][2] = t[i][2] - t[i][1];
        e[i][3] = t[i][1] - t[i][3];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *reinterpret_cast<uint32_t*>(s_v + (((i * 4 + j) * W_TILES + tt) * W_CK + ((cp * 2) ^ (((tt >> 3) & 1) << 3))) * 2) = __builtin_bit_cast(uint32_t, e[i][j]);
      return;
    }
    float d[2][4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t pk = *reinterpret_cast<const uint32_t*>(s_raw + ((2 * tty + i) * W_PW + ((2 * ttx + j) ^ (((2 * ttx + j) >> 2) & 1))) * (W_CK * 2) + cp * 4);
        d[0][i][j] = ld16<EK>(pk);
        d[1][i][j] = ld16<EK>(pk >> 16);
      }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float t[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        t[0][j] = d[c][0][j] - d[c][2][j];
        t[1][j] = d[c][1][j] + d[c][2][j];
        t[2][j] = d[c][2][j] - d[c][1][j];
        t[3][j] = d[c][1][j] - d[c][3][j];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        d[c][i][0] = t[i][0] - t[i][2];
        d[c][i][1] = t[i][1] + t[i][2];
        d[c][i][2] = t[i][2] - t[i][1];
        d[c][i][3] = t[i][1] - t[i][3];
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        *reinterpret_cast<uint32_t*>(s_v + (((i * 4 + j) * W_TILES + tt) * W_CK + ((cp * 2) ^ (((tt >> 3) & 1) << 3))) * 2) = pack2<EK>(d[0][i][j], d[1][i][j]);
  };

  if (use_dma) dma_u(0, 0);
  gload(0);
  if constexpr (PRO != PRO_RAW) __syncthreads();                // the prologue table is in LDS
  lstore(0);
  if (use_dma) DD_WAIT_VM(0);  // this wave's pieces of U(0) have landed
  __syncthreads();
  gload(1);
  transform(0);
  __syncthreads();
#pragma unroll 1
  for (int chunk = 0; chunk < NCHUNK; ++chunk) {
    const int cur = chunk & 1, nxt = cur ^ 1;
    const bool more = chunk + 1 < NCHUNK;
    if (more) lstore(nxt);                                      // registers hold chunk + 1
    __syncthreads();                                            // A: raw (and, without DMA, U) of chunk + 1 visible
    if (chunk + 2 < NCHUNK) gload(chunk + 2);
    // The DMA goes BEHIND every point where hipcc waits for its own loads (lstore above, and the conservative vmcnt waits it puts behind
    // the barrier): s_waitcnt counts ALL outstanding VMEM operations, so a compiler wait with uncounted DMAs in flight would wait for them.
    // U[nxt] was last read by the MFMAs of chunk - 1, two barriers ago; the DMA has the MFMAs + the transform below to land.
    if (more && use_dma) dma_u(chunk + 1, nxt);
    {
      const char* s_v = smem + W2_V_OFF + cur * W_V_BYTES;
      const char* s_u = smem + W2_U_OFF + cur * W_U_BYTES;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int pos = 2 * wave + a;
        uint4 wf[2], vf[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) wf[m] = *reinterpret_cast<const uint4*>(s_u + ((pos * W_NT + m * 32 + li) * W_CK + ((g ^ ((li >> 3) & 1)) * 8)) * 2);
#pragma unroll
        for (int n = 0; n < 2; ++n) vf[n] = *reinterpret_cast<const uint4*>(s_v + ((pos * W_TILES + n * 32 + li) * W_CK + ((g ^ ((li >> 3) & 1)) * 8)) * 2);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) mma_step<EK>(acc[a][m][n], wf[m], vf[n]);
      }
    }
    if (more) transform(nxt);                                   // VALU + LDS while the matrix pipe drains the 8 MFMAs above
    if (use_dma) DD_WAIT_VM(0);  // the DMA of U(chunk + 1) (and the register prefetch behind it) landed
    __syncthreads();                                            // B: V (DMA: and U) of chunk + 1 complete, fragment reads of this chunk done
  }

  // ---- epilogue: as v1, plus the GroupNorm partial sums of this layer's outputs (STATS) ------------------------------------------------
  float* s_m = reinterpret_cast<float*>(smem);
  char* out_b = reinterpret_cast<char*>(p.out) + (size_t)b * HW * COUT * 2;
  // local statistics slots: COUT == 256: the workgroup's 64 couts are exactly GroupNorm group `nsplit` -> slot 0 only;
  //                         COUT == 64: four groups of 16 couts -> slot = (32 m + 8 cg) / 16
  float ls[4] = {0.f, 0.f, 0.f, 0.f}, lq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int blk = 0; blk < 4; ++blk) {
    const int m = blk >> 1, n = blk & 1;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int pos = 2 * wave + a;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 v;
        if (m == 0 && n == 0) v = make_float4(acc[a][0][0][q * 4], acc[a][0][0][q * 4 + 1], acc[a][0][0][q * 4 + 2], acc[a][0][0][q * 4 + 3]);
        else if (m == 0) v = make_float4(acc[a][0][1][q * 4], acc[a][0][1][q * 4 + 1], acc[a][0][1][q * 4 + 2], acc[a][0][1][q * 4 + 3]);
        else if (n == 0) v = make_float4(acc[a][1][0][q * 4], acc[a][1][0][q * 4 + 1], acc[a][1][0][q * 4 + 2], acc[a][1][0][q * 4 + 3]);
        else v = make_float4(acc[a][1][1][q * 4], acc[a][1][1][q * 4 + 1], acc[a][1][1][q * 4 + 2], acc[a][1][1][q * 4 + 3]);
        *reinterpret_cast<float4*>(s_m + ((pos * 32 + li) * 32 + 8 * q + 4 * g)) = v;
      }
    }
    __syncthreads();
    {
      // all 512 lanes: lane -> (tile tj, cout quad cg*8 + c8*4 .. +3, output row dy); it reads the three M rows its output row needs
      // (dy = 0: rows 0,1,2 of A^T M; dy = 1: rows 1,2,3), forms both output columns and stores 2 pixels x 4 couts
      const int tj = tid >> 4, cg = (tid >> 2) & 3, c8 = (tid >> 1) & 1, dy = tid & 1;
      const int T = n * 32 + tj;
      const int ty = T / W_TX, tx = T - ty * W_TX;
      const int co = nsplit * W_NT + m * 32 + cg * 8 + c8 * 4;
      const int slot = (COUT == 256) ? 0 : ((m * 32 + cg * 8) >> 4);
      float t[4][4];                                            // (A^T M)[dy][j][cout]
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 r1 = *reinterpret_cast<const float4*>(s_m + (((1 * 4 + j) * 32 + tj) * 32 + cg * 8 + c8 * 4));
        const float4 r2 = *reinterpret_cast<const float4*>(s_m + (((2 * 4 + j) * 32 + tj) * 32 + cg * 8 + c8 * 4));
        const float4 re = *reinterpret_cast<const float4*>(s_m + ((((dy ? 3 : 0) * 4 + j) * 32 + tj) * 32 + cg * 8 + c8 * 4));
        if (dy == 0) { t[j][0] = re.x + r1.x + r2.x; t[j][1] = re.y + r1.y + r2.y; t[j][2] = re.z + r1.z + r2.z; t[j][3] = re.w + r1.w + r2.w; }
        else { t[j][0] = r1.x - r2.x - re.x; t[j][1] = r1.y - r2.y - re.y; t[j][2] = r1.z - r2.z - re.z; t[j][3] = r1.w - r2.w - re.w; }
      }
      const float4 b4 = *reinterpret_cast<const float4*>(p.bias + co);
      const float bias[4] = {b4.x, b4.y, b4.z, b4.w};
      float ssum = 0.f, ssq = 0.f;
      const int gy = y0 + 2 * ty + dy;
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        const int gx = x0 + 2 * tx + dx;
        if (gy < h && gx < w) {
          float v[4];
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] = (dx == 0 ? (t[0][c] + t[1][c] + t[2][c]) : (t[1][c] - t[2][c] - t[3][c])) + bias[c];
          if constexpr (STATS) {                                // from the fp32 values, before the 16-bit rounding (as dd_igemm2.hip)
            ssum += (v[0] + v[1]) + (v[2] + v[3]);
            ssq += fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], v[3] * v[3])));
          }
          *reinterpret_cast<uint2*>(out_b + act_offset(COUT, h, w, 0, co, gy, gx) * 2) = make_uint2(pack2<EK>(v[0], v[1]), pack2<EK>(v[2], v[3]));
        }
      }
      if constexpr (STATS) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (k == slot) { ls[k] += ssum; lq[k] += ssq; }
      }
    }
    __syncthreads();
  }
  if constexpr (STATS) {
    // every wave holds partial sums: butterfly inside each wave (fp32), fp64 from there on
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
      for (int k = 0; k < 4; ++k) { ls[k] += __shfl_xor(ls[k], off, 64); lq[k] += __shfl_xor(lq[k], off, 64); }
    double* s_red = reinterpret_cast<double*>(smem);            // every LDS image is dead behind the loop's last barrier
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k) { s_red[wave * 8 + 2 * k] = (double)ls[k]; s_red[wave * 8 + 2 * k + 1] = (double)lq[k]; }
    }
    __syncthreads();
    constexpr int NG_LOCAL = (COUT == 256) ? 1 : 4;
    if (tid < 2 * NG_LOCAL) {
      double tot = 0.0;
#pragma unroll
      for (int wv = 0; wv < W_THREADS / 64; ++wv) tot += s_red[wv * 8 + tid];
      const int gbase = (COUT == 256) ? nsplit : 0;             // COUT == 256: this workgroup's couts are group `nsplit`
      atomicAdd(p.stats_out + ((size_t)b * STAT_SLOTS + (wgid % STAT_SLOTS)) * STAT_STRIDE + gbase * 2 + tid, tot);
    }
  }
}

// (a, b, e) of the prologue for every (image, channel): a = gamma / sqrt(var + eps), b = beta - mean * a from the producing layer's
// GroupNorm partial sums (32 slots x 4 groups x (Sigma, Sigma x^2), fp64), e = E[t][c] (PRO_GN_ADD) or 0.  One block per image.
__global__ void __launch_bounds__(256) wino_gn_table_kernel(const double* __restrict__ stats, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const float* __restrict__ emb,
                                                            const long long* __restrict__ tvec, int t_base, int t_bstride,
                                                            float* __restrict__ tab, int C, double count_per_group) {
  const int b = blockIdx.x, c = threadIdx.x;
  if (c >= C) return;
  const int grp = c / (C / GN_GROUPS);
  double sm = 0.0, sq = 0.0;
  for (int sl = 0; sl < STAT_SLOTS; ++sl) {
    const double* st = stats + ((size_t)b * STAT_SLOTS + sl) * STAT_STRIDE + grp * 2;
    sm += st[0];
    sq += st[1];
  }
  const double mean = sm / count_per_group;
  double var = sq / count_per_group - mean * mean;              // biased, as torch
  var = var > 0.0 ? var : 0.0;
  const double a = (double)gamma[c] / sqrt(var + (double)GN_EPS);
  float e = 0.f;
  if (emb != nullptr) e = emb[(size_t)clamp_t(tvec[t_base + b * t_bstride]) * COND_C + c];
  reinterpret_cast<float4*>(tab)[(size_t)b * C + c] = make_float4((float)a, (float)((double)beta[c] - mean * a), e, 0.f);
}

hipError_t launch_wino_gn_table(const ConvParams& p, int C, float* tab, bool with_emb, hipStream_t s) {
  hipLaunchKernelGGL(wino_gn_table_kernel, dim3((unsigned)p.B), dim3(256), 0, s, p.stats_in, p.gn_gamma, p.gn_beta,
                     with_emb ? p.emb : nullptr, p.tvec, p.t_base, p.t_bstride, tab, C, (double)p.h * (double)p.w * (double)(C / GN_GROUPS));
  return hipGetLastError();
}

namespace {
template <int EK, bool PK, int CIN, int COUT, int PRO, bool STATS>
hipError_t launch_wino_inst(const ConvParams& p, hipStream_t s) {
  constexpr int SMEM = W2_SMEM + 3 * CIN * 4;                   // + the prologue table
  static_assert(SMEM <= 160 * 1024, "LDS");
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wino_raw_v2_kernel<EK, PK, CIN, COUT, PRO, STATS>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  ConvParams q = p;
  q.tiles_x = (p.w + W_TW - 1) / W_TW;
  q.tiles_y = (p.h + W_TH - 1) / W_TH;
  const unsigned n_wg = (unsigned)(q.tiles_x * q.tiles_y * q.B * (COUT / W_NT));
  hipLaunchKernelGGL((conv_wino_raw_v2_kernel<EK, PK, CIN, COUT, PRO, STATS>), dim3(n_wg), dim3(W_THREADS), SMEM, s, q);
  return hipGetLastError();
}
template <int EK, bool PK>
hipError_t launch_wino_layer_ek(int layer, const ConvParams& p, hipStream_t s) {
  switch (layer) {
    case 2: return launch_wino_inst<EK, PK, HID_C, COND_C, PRO_GN, true>(p, s);          // conv2: relu(gn1(y1)) -> y2 (+ GN2 statistics)
    case 3: return launch_wino_inst<EK, PK, COND_C, HID_C, PRO_GN_ADD, true>(p, s);      // conv3: relu(gn2(y2)) + cond + E[t] -> y3 (+ GN3)
    case 5: return launch_wino_inst<EK, PK, COND_C, COND_C, PRO_GN_ADD, false>(p, s);    // Swin convA
    case 6: return launch_wino_inst<EK, PK, COND_C, COND_C, PRO_RAW, false>(p, s);       // Swin convB
    case 7: return launch_wino_inst<EK, PK, COND_C, HID_C, PRO_RAW, true>(p, s);         // Swin pred.0 (+ GN3 statistics)
    default: return hipErrorInvalidValue;
  }
}
}  // namespace

bool conv_wino_supports(int layer) { return layer == 2 || layer == 3 || layer == 5 || layer == 6 || layer == 7; }

// p as dd_api.cpp fills it for the direct kernel of `layer`, except p.wpack = the layer's wino_pack_u image and (layers 2, 3, 5)
// p.wino_tab = the [B][CIN][4] table written by launch_wino_gn_table on the same stream just before
hipError_t launch_conv_wino_layer(int layer, int ek, const ConvParams& p, hipStream_t s, bool packed_f16_transform) {
  if (ek == EK_BF16) return launch_wino_layer_ek<EK_BF16, false>(layer, p, s);
  if (ek != EK_F16) return hipErrorInvalidValue;
  return packed_f16_transform ? launch_wino_layer_ek<EK_F16, true>(layer, p, s) : launch_wino_layer_ek<EK_F16, false>(layer, p, s);
}

hipError_t launch_conv_wino_raw(int ek, const ConvParams& p, hipStream_t s, int version) {
  if (ek != EK_BF16 && ek != EK_F16) return hipErrorInvalidValue;
  ConvParams q = p;
  q.tiles_x = (p.w + W_TW - 1) / W_TW;
  q.tiles_y = (p.h + W_TH - 1) / W_TH;
  if (version >= 2) {
    return launch_conv_wino_layer(6, ek, p, s, version == 3);
  }
  static bool attr_set[2] = {false, false};
  const void* fn = ek == EK_BF16 ? reinterpret_cast<const void*>(&conv_wino_raw_kernel<EK_BF16>)
                                 : reinterpret_cast<const void*>(&conv_wino_raw_kernel<EK_F16>);
  if (!attr_set[ek == EK_F16]) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, W_SMEM);
    if (e != hipSuccess) return e;
    attr_set[ek == EK_F16] = true;
  }
  const unsigned n_wg = (unsigned)(q.tiles_x * q.tiles_y * q.B * (COND_C / W_NT));
  if (ek == EK_BF16) hipLaunchKernelGGL(conv_wino_raw_kernel<EK_BF16>, dim3(n_wg), dim3(W_THREADS), W_SMEM, s, q);
  else hipLaunchKernelGGL(conv_wino_raw_kernel<EK_F16>, dim3(n_wg), dim3(W_THREADS), W_SMEM, s, q);
  return hipGetLastError();
}

// Packed weights the kernel streams: [cout split (COUT / 64)][chunk (CIN / 16)][position 4*xi + nu][co 64][ch 16] of U = G g G^T
// (computed in double, rounded once to the operand type by the caller-supplied converter).
size_t wino_pack_bytes(int cout, int cin) { return (size_t)cout * cin * 16 * 2; }
void wino_pack_u(const float* w_oihw, int cout, int cin, uint16_t (*cvt)(float), uint16_t* out) {
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  const int nsplit = cout / W_NT, nchunk = cin / W_CK;
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci) {
      const float* g = w_oihw + ((size_t)co * cin + ci) * 9;
      double t[4][3];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 3; ++j) t[i][j] = G[i][0] * g[0 * 3 + j] + G[i][1] * g[1 * 3 + j] + G[i][2] * g[2 * 3 + j];
      for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
          const double u = t[i][0] * G[j][0] + t[i][1] * G[j][1] + t[i][2] * G[j][2];
          const int sp = co / W_NT, cl = co % W_NT, ch = ci / W_CK, ck = ci % W_CK;
          out[((((size_t)sp * nchunk + ch) * 16 + (i * 4 + j)) * W_NT + cl) * W_CK + ck] = cvt((float)u);
        }
    }
  (void)nsplit;
}

}  // namespace dd
