// dd_kernels.h -- device-side parameter blocks and host launchers shared by the C-ABI layer
// (dd_api.cpp) and the kernel translation units (dd_igemm2.hip, dd_naive.hip, dd_misc.hip, dd_bwd.hip, dd_wgrad.hip).
//
// Internal data layout in HBM (DESIGN.md "Data layout"):
//   activations  NHWC  [B][h][w][C]   element type per precision mode (bf16 / f16 / f32)
//   state x_t    NHWC  [B][h][w][16]  fp32, ping-pong pair
//   GroupNorm statistics  [B][DD_STAT_SLOTS][DD_STAT_STRIDE] doubles: per slot (sum, sumsq) x 4 groups
//   packed conv weights   [n_tile][cin_chunk][tap_group][tap][NT][CK] elements (see pack_conv_weights)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dd {

constexpr int LATENT_C = 16;   // depth_feature_dim   (reference src/model/diffusion_dcbase_model.py:84)
constexpr int HID_C = 64;      // hidden width        (reference src/model/head/ddim_depth_estimate_res.py:303,315)
constexpr int COND_C = 256;    // channels_in/fpn_dim (reference ...res.py:27-28)
constexpr int GN_GROUPS = 4;   // nn.GroupNorm(4, C)  (reference ...res.py:305,309,317,321)
constexpr int EMB_ROWS = 1280; // nn.Embedding(1280, 256) (reference ...res.py:313)
// timestep -> row of the time embedding / etab tables.  The reference's nn.Embedding raises on an out-of-range index; a device
// kernel cannot, so rows are clamped (dd_set_schedule rejects num_train_timesteps > EMB_ROWS, so loop timesteps never clamp).
__host__ __device__ inline long long clamp_t(long long t) { return t < 0 ? 0 : (t >= EMB_ROWS ? EMB_ROWS - 1 : t); }
constexpr float GN_EPS = 1e-5f;
constexpr float BN_EPS = 1e-5f;

// GroupNorm partial sums are spread over DD_STAT_SLOTS cache lines per sample so that the one
// atomic per workgroup per group does not serialise on a single L2 channel.
constexpr int STAT_SLOTS = 32;
constexpr int STAT_STRIDE = 16;   // doubles per slot (128 B): [g*2+0]=sum, [g*2+1]=sumsq, g<4; rest pad

// Element kinds.  EK_F32 / EK_BF16 / EK_F16 name both a storage type and an MFMA operand type.  EK_BF16M is a MODE of the fused path, not a
// type: "bf16 operands, f16 storage" -- the two large convolutions (conv2 / conv3 and the Swin fuse convs: 94 % of the FLOPs) contract
// bf16 operands on v_mfma_f32_32x32x16_bf16, every tensor that crosses a kernel boundary in front of a GroupNorm (y1, y2, y3) and the
// condition map are STORED as f16 (same bytes, 11-bit mantissa; GroupNorm bounds their range), and the two thin layers conv1 / conv4
// (16 <-> 64 channels, 6 % of the FLOPs) and the once-per-image conv3(cond) run as f16 kernels.  Why: DESIGN.md section 4 (the error
// budget of tools/bf16_error_budget.py: pure bf16 sits at 1.2e-3 depth RMSE, above the 1e-3 tolerance; this mode at ~0.4x of it).
// EK_F16S is the second MODE: "split f16" (DD_PREC_F16X3) -- every tensor between kernels is fp32 (the layouts of the fp32 mode), every MFMA
// operand is carried as an f16 PAIR hi + lo (hi = f16(v), lo = f16(v - hi): ~22 mantissa bits) and a product W.P is contracted as
// Whi.Phi + Whi.Plo + Wlo.Phi on v_mfma_f32_32x32x16_f16 (three MFMAs; the dropped Wlo.Plo term is 2^-22 relative), fp32 accumulation.
// Operands are pre-scaled by exact powers of two (weights x SPLIT_WSCALE at pack time; the patch x SPLIT_PSCALE in the prologues that sit
// behind a GroupNorm, whose output is bounded -- the unbounded inputs, the state x of conv1 and the raw tensors the Swin convB / pred.0 read,
// are carried unscaled: f16 overflows only beyond |v| = 65504) so that the lo halves of ordinary values stay in f16's normal range, and the
// epilogue multiplies the accumulators by the exact inverse.  (gfx950's f16 MFMA honours subnormal inputs -- tools/micro/f16_denorm_probe.hip,
// profiles/history/r03_run1_f16_denorm_probe.txt -- so the scaling buys bits for small values, it is not needed for correctness.)  It is the
// abs-1e-3-on-depth mode at ~1/3 of the 16-bit MFMA rate (5x the fp32-operand MFMA rate): DESIGN.md section 4.
// EK_F16R is the third MODE: "refined f16" (DD_PREC_F16R; round 4) -- the 16-bit mode that holds the depth tolerance at KITTI's depth range.
// tools/bf16_error_budget.py --log-scale 1.8 shows that the f16 mode's depth error (1.0e-3 RMSE at 0.5..60 m) is NOT made by the two large
// convolutions (their f16 operands and weights: 2.1e-4 / 3.2e-4) but by what surrounds them: the once-per-image conv3(cond) on f16 operands
// (4.9e-4) and its f16 storage (3.6e-4), conv4's weights (4.4e-4) and operand (3.3e-4), y3's f16 storage (3.5e-4), conv1's weights
// (2.4e-4).  Those are 6 % of the FLOPs and HBM-bound, so they are made (near-)exact where that is (nearly) free:
//   conv1            its WEIGHTS as an f16 pair hi + lo (two MFMAs on 3 % of the FLOPs; the state x stays a single f16 operand: 1.5e-4)
//   conv2, conv3     the f16 kernels' arithmetic, unchanged: single MFMA per product
//   conv3(cond)      once per image on the split-f16 kernel (EK_F16S) from the fp32 condition map, then reformatted (launch_cadd_reformat)
//   hand-overs       y3 and that hoisted term travel as BLOCK-SCALED INT16 (option "f16r_wide" = 1): y3 as int16 with one fp32 scale per PIXEL
//                    (the maximum |.| over its 64 channels: conv3's epilogue has a pixel's channels in the two lanes li / li + 32, conv4's
//                    staging item is 8 channels of one pixel), the hoisted term as int16 with one scale per accumulator block (32 pixels x 32
//                    couts: a wave-uniform scalar in conv3's accumulator initialisation).  f16's bytes at ~15 bits relative to the block's largest
//                    value: 4e-5 instead of f16's 3.5e-4 depth RMSE per tensor (the emulator; an fp32 hand-over measured the same error at
//                    conv3 +26 us / conv4 +15 us per step and was dropped: profiles/history/r04_call1_*, r04_call2_*)
//   conv4            weights as an f16 pair hi + lo IN ONE MFMA: the 32 x 32 x 16 instruction has 32 cout rows and conv4 16 couts, so rows
//                    16..31 of the A operand -- zero padding until now -- carry lo * 2^11 and the epilogue adds the two accumulator halves
//                    (lane-local: register quads q and q + 2); option "f16r_p4": the operand relu(gn3(y3)) as a pair as well (two MFMAs)
// Instantiated (dd_igemm2_cfg.h) for the layers whose kernel differs from the f16 mode's: conv1 and the hoisted conv3.  Forward only, Res denoiser;
// DESIGN.md section 4.
enum ElemKind : int { EK_F32 = 0, EK_BF16 = 1, EK_F16 = 2, EK_BF16M = 3, EK_F16S = 4, EK_F16R = 5 };
constexpr float Q15_ONE = 32767.f;
__host__ __device__ constexpr int opnd_kind(int ek) { return ek == EK_BF16M ? (int)EK_BF16 : (ek == EK_F16S || ek == EK_F16R) ? (int)EK_F16 : ek; }    // MFMA operands of the large convolutions / gradients
__host__ __device__ constexpr int store_kind(int ek) { return (ek == EK_BF16M || ek == EK_F16R) ? (int)EK_F16 : ek == EK_F16S ? (int)EK_F32 : ek; }    // y1 / y2 (/ y3) in HBM
__host__ __device__ constexpr int cond_kind(int ek) { return ek == EK_F16R ? (int)EK_F32 : store_kind(ek); }       // the condition map at latent size (EK_F16R: fp32, read once per image by the split conv3(cond))
constexpr float STACK_LSCALE = 2048.f;    // 2^11: conv4's stacked weight image (EK_F16R): rows 16..31 = f16((w - f16(w)) * 2^11), the lo half at full f16 precision
constexpr float SPLIT_WSCALE = 256.f;     // 2^8: default-initialised 3x3 weights (|w| <= 1/sqrt(9 Cin) ~ 0.02..0.08) land at ~5..20, their lo halves at ~2^-9
constexpr float SPLIT_PSCALE = 16.f;      // 2^4: activations behind a GroupNorm + ReLU (+ condition map + E[t]); overflow only beyond |v| = 4094
__host__ __device__ constexpr float split_pscale(int pro) { return (pro == 1 /* PRO_GN */ || pro == 2 /* PRO_GN_ADD */) ? SPLIT_PSCALE : 1.f; }
__host__ __device__ constexpr float split_oscale(int pro) { return 1.f / (SPLIT_WSCALE * split_pscale(pro)); }     // exact powers of two

enum Prologue : int {
  PRO_X = 0,       // conv1: input = DDIM-updated state (c1*x + c2*relu(gn4(y4))), also written back
  PRO_GN = 1,      // input = relu(gn(y))
  PRO_GN_ADD = 2,  // conv3 / Swin convA: input = relu(gn(y)) + cond + E[t]
  PRO_RAW = 3      // Swin convB / pred.0: input is used as stored (no normalisation in between)
};

// One 3x3 convolution launch of the fused path.
struct ConvParams {
  const void* in;           // producing layer's raw output (NHWC, activation element type); PRO_X: x (fp32)
  const void* wpack;        // packed weights (activation element type)
  const float* bias;        // [COUT padded to NT multiple]
  void* out;                // raw conv output + bias (pre-GroupNorm), NHWC
  double* stats_out;        // [B][STAT_SLOTS][STAT_STRIDE] accumulators of THIS layer's GroupNorm
  const double* stats_in;   // statistics the prologue normalises with (PRO_X: GN4 of the previous step)
  const float* gn_gamma;    // affine of the GroupNorm applied in the prologue
  const float* gn_beta;
  const void* cond;         // PRO_GN_ADD: condition map NHWC [B][h][w][256], activation element type
  const float* emb;         // PRO_GN_ADD: time-embedding table [1280][256] fp32
  const long long* tvec;    // PRO_GN_ADD: timestep(s); t = tvec[t_base + b * t_bstride]
  int t_base, t_bstride;
  long long t_known = -1;   // >= 0: the timestep of every image, known on the host when the launch was recorded (the DDIM loop: the schedule's
                            // steps are fixed per plan) -- spares the kernel the load in front of its embedding / E[t] table addresses; -1: read tvec
  const float* y4;          // PRO_X: raw conv4 output of the previous step, fp32 NHWC [B][h][w][16]
  float* xout;              // PRO_X: updated state written here (interior pixels of each tile)
  const float* c1c2;        // PRO_X: [T][2] DDIM coefficients; entry (step-1) is applied when step > 0
  int step;                 // PRO_X: 0 = no update (x is used as is)
  int B, h, w;
  int tiles_x, tiles_y;
  // conv3 with the condition term hoisted out of the loop (layer 9): out += cadd[pixel][cout] + sum over the taps that
  // fall inside the image of etab[t][tap][cout]   (conv is linear: conv3(r + cond + E[t]) = conv3(r) + conv3(cond) + conv3(E[t]))
  const float* cadd;        // conv3(cond) without bias, fp32, activation layout [B][2][h][w][32]
  const float* cadd_scale;  // EK_F16R: cadd holds int16 quads; value = int16 * cadd_scale[block], block = ((tile * WAVES + wave) * WN + n) * WM + m
  float* out_scale;         // EK_F16R conv3: y3 is written as int16; out_scale[b * h * w + pixel] = that pixel's max |.| / 32767
  const float* etab;        // [EMB_ROWS][10][64] fp32: per-tap W3_tap . E[t] (entries 0..8) and their sum (entry 9)
  // Swin denoiser with the step-invariant part of pred.0(convB(convA(.))) hoisted (kernel id SWIN_PRED_H): this step's rows of the
  // time-embedding table [SWIN_TT_ROWS][64] fp32 -- row 0 is added to every pixel, row 1 + 7 r + c to the pixels of border class (r, c)
  const float* ttab;
  int ttab_bstride;         // floats between the tables of consecutive images: 0 in the loop (one timestep for the whole batch), SWIN_TT_ROWS * 64 for a single call with per-sample timesteps
  const float* bcorr;       // SWIN_PRED5_H: [B][swin_ring_stride][64] fp32, subtracted at the pixels on the image border (corners: ring entry + corner entry)
  const void* addend;       // FPN lateral convs (layers 10..13): optional top-down term added after the ReLU (activation layout), or NULL
  // HAHI neck layers (30..41): the input / output tensor is a channel range of a wider channel-blocked buffer (the concatenation the
  // fusion conv reads): buffer width and first channel, both multiples of 32; 0 = the tensor is the whole buffer (every other layer)
  int in_cstride, in_coff, out_cstride, out_coff;
  unsigned long long* prof; // -DDD_PHASE_PROF=1 builds only (tools/phase_prof.py): 8 x u64 per workgroup: wall-clock (100 MHz) at kernel entry,
                            // GroupNorm table done, first patch + weights in LDS, main loop done, stores issued, exit; [6] = HW_ID, [7] = XCC_ID
  int persist_slots;        // persistent kernels (dd_thin.hip): resident workgroup slots to fill (0 = 512: two per CU on the 256 CUs)
  int ablate;               // TIMING EXPERIMENTS ONLY (results are wrong when non-zero): bit0 skip in-loop patch transform,
                            // bit1 skip in-loop patch loads, bit2 skip in-loop weight DMA, bit3 skip MFMAs, bit4 skip output
                            // stores, bit5 skip GroupNorm statistics, bit6 skip the per-stage barrier
};

// ---- fused implicit-GEMM path (dd_igemm2.hip) ------------------------------------------------
// layer ids: see dd_igemm2_cfg.h; ek: element kind.  Weights are packed with the LDS swizzle pre-applied
// (16-B piece j of block row r is stored at j ^ ((r / (256/rowbytes)) & (rowbytes/16 - 1))).
// Kernel ids of the BIG-TILE forms of the hoisted conv3 pair (dd_igemm2_cfg.h): layer 8 (conv3(cond), once per image) and layer 9 (conv3 in
// the loop) on 16x32-pixel tiles -- four waves of 128 pixels x 64 couts each, 0.75 instead of 1.0 LDS fragment reads per MFMA, half the
// weight stream per pixel -- chosen per launch when there are more 8x32 tiles than resident workgroup slots (the two must agree: layer 8
// leaves its result in the accumulator-fragment order of layer 9's tiles).  Same packed weights as layers 8 / 9.
constexpr int BIG_CONV3C = 48, BIG_CONV3H = 49;
// layer 9 on its 8x32 tiles with ONE patch buffer (dd_igemm2_cfg.h, ONEBUF): the next chunk's patch goes into the buffer the MFMAs just read, behind a
// second workgroup barrier per stage; 52 KB of LDS = three workgroups per CU.  Same tiles, packed weights and accumulator-fragment order as layer 9.
constexpr int ONE_CONV3H = 46;
// layer 8 (the once-per-image conv3(cond)) reading the caller's NCHW fp32 condition tensor DIRECTLY: a staging item's eight channels are eight
// 4-byte loads from eight channel planes (consecutive lanes = consecutive pixels of one plane: whole 128-byte segments) instead of two 16-byte
// loads from the channel-blocked copy -- the copy (438 MB read + 438 MB written per four KITTI maps: 189 us of a 7.5-ms step in the refined f16
// mode) is not made at all.  Split-f16 kernel only (the refined mode's hoisted plans with an explicit condition tensor); same tiles, packed
// weights, arithmetic and output order as layer 8 in that kind -- bit-identical results.
constexpr int CONV3C_NCHW = 47;
// Swin / MPViT denoiser, forward-only plans: upsample_fuse (convA, convB: no norm, no activation) and pred.0 are ONE linear map of
// s = up(feat) + E[t] + NE(x_t) (reference ...swin_addHAHI.py:321-333,378-380), so
//   pred.0(convB(convA(s))) = W3*WB*WA*NE(x_t)  +  [W3*(WB*(WA*up(feat) + a) + b)]  +  W3*WB*WA*(E[t] on every pixel)  + b3
// with every convolution zero-padding its own input as the reference's does.  The bracket is computed ONCE per image (layer 6 kernel on
// convA's and convB's weights, then layer 8: accumulator-fragment order, as the Res variant's hoisted conv3(cond)); the E[t] term is constant
// over the image except within three pixels of its border: a table per loop step with one row per border class (swin_ttab, dd_misc.hip).
//   SWIN_CONVA_H = convA on relu(gn2(y2)) alone (layer 5 without the condition / embedding addends, no bias)
//   SWIN_PRED_H  = pred.0 (layer 7) whose accumulators start at the hoisted term and whose epilogue adds the table rows
constexpr int SWIN_CONVA_H = 50, SWIN_PRED_H = 52;
//   SWIN_PRED5_H = pred.0 and convB as ONE 5x5 convolution 256 -> 64 on convA's result (W5[u] = sum over e + d = u of W3[e] . WB[d], built per
//   parameter generation by swin_compose, dd_misc.hip): 0.82 instead of 1.47 MFLOP per pixel and step and no convB result in HBM.  The 5x5
//   form also sums, at the pixels ON the image border, the terms W3[e] . convB(.)(q + e) for taps e that leave the image -- which the
//   reference's pred.0 zero-pads away; they are computed per step from the border rows / columns of convA's result (swin_bcorr: ring
//   buffer ConvParams::bcorr) and subtracted in the epilogue.  Accumulator start values and E[t] rows as SWIN_PRED_H.
constexpr int SWIN_PRED5_H = 53;
// the same on 16x32-pixel tiles (the tiling of BIG_CONV3C / BIG_CONV3H: half the weight stream per pixel, 40 MFMAs per stage, 0.75 LDS reads per
// MFMA), picked with the same rule (plan_big_tiles, dd_api.cpp); its accumulator start values come from BIG_CONV3C
constexpr int SWIN_PRED5B_H = 51;
// ring buffer of the border pixels of an h x w image: top row, bottom row, left column, right column (without the corners)
__host__ __device__ inline int swin_ring_size(int h, int w) { return 2 * w + 2 * (h > 2 ? h - 2 : 0); }
__host__ __device__ inline int swin_ring_stride(int h, int w) { return swin_ring_size(h, w) + 4; }      // + one entry per corner: its sideways taps (swin_bcorr_line_kernel)
__host__ __device__ inline int swin_ring_index(int y, int x, int h, int w) {
  return y == 0 ? x : (y == h - 1 ? w + x : (x == 0 ? 2 * w + (y - 1) : 2 * w + (h - 2) + (y - 1)));
}
// border classes of the E[t] term along one axis of n pixels: position y < 3 -> y, y >= n - 3 -> R - (n - y), else 3, with R = min(n, 7):
// the value at a pixel depends only on its distances (capped at 3) to the two borders, i.e. equals the value at pixel class(y) of an R-pixel axis
constexpr int SWIN_TT_AX = 7, SWIN_TT_ROWS = 1 + SWIN_TT_AX * SWIN_TT_AX;
__host__ __device__ inline int swin_tt_class(int y, int n) { const int R = n < SWIN_TT_AX ? n : SWIN_TT_AX; return y < 3 ? y : (y >= n - 3 ? R - (n - y) : 3); }
__host__ __device__ inline int swin_tt_ref(int n) { return n - 1 < 3 ? n - 1 : 3; }     // the class whose value row 0 carries
// workgroups of a persistent launch (dd_thin.hip): B x n with n workgroups per image, at most `slots` in all
inline int persist_grid(int B, int tiles_per_img, int slots) {
  int n = slots / (B > 0 ? B : 1);
  if (n < 1) n = 1;
  if (n > tiles_per_img) n = tiles_per_img;
  return B * n;
}
struct PackGeom { int cin, cout, cout_pad, ck, tg, nt, th, ks, planes, stack; };   // th = output tile height (tile width is 32), ks = kernel size,
                                                                            // planes = 2: split f16 image, every stage block = [hi | lo];
                                                                            // stack = 1 (conv4, EK_F16R): cout rows cout..2 cout-1 = the lo halves times STACK_LSCALE
hipError_t launch_conv_igemm2(int layer, int ek, const ConvParams& p, hipStream_t s);
// conv4 (64 -> 16) as a persistent streaming kernel (dd_thin.hip): same ConvParams and packed weights as layer 4; ek = EK_F16 / EK_BF16
//   stack = the stacked hi / lo weight image (EK_F16R); in_q15 = y3 arrives as int16 with a per-pixel scale (ConvParams::cadd_scale = that scale
//   array here) instead of the 2-byte kind; psplit = the operand as an f16 pair too (stack only)
hipError_t launch_conv4_stream(int ek, const ConvParams& p, hipStream_t s, bool stack = false, bool in_q15 = false, bool psplit = false);
// EK_F16R: the once-per-image term conv3(cond), left by the split-f16 layer 8 in the accumulator-fragment order of 8x32 tiles (fp32), into
// the order / element type the loop's conv3 reads: 8x32 or 16x32 tiles (big); out_kind: 1 = f16 quads, 2 = int16 quads + one fp32 scale per
// (tile, wave, n, m) block of 32 pixels x 32 couts (scales) (dd_misc.hip)
hipError_t launch_cadd_reformat(const float* src, void* dst, float* scales, int B, int h, int w, int big, int out_kind, hipStream_t s);
PackGeom conv_pack_geom2(int layer, int ek);

// ---- weights into kernel layout on the device (dd_misc.hip): the packed image of pack_conv_weights() in dd_api.cpp, bit for bit ----
size_t pack_weights_bytes(const PackGeom& g, int ek);
hipError_t launch_pack_weights(const float* src_oihw, void* dst, const PackGeom& g, int ek, bool swizzle, bool transposed, hipStream_t s);
hipError_t launch_transpose_flip(const float* w, float* wt, int cout, int cin, int kk, hipStream_t s);
// *out_bits = max(*out_bits, bits of max |w|) (non-negative floats order like their bit patterns; NaN sorts above every finite value)
hipError_t launch_max_abs(const float* w, long long n, unsigned* out_bits, hipStream_t s);
hipError_t launch_spin(long long ticks_100mhz, long long* stamp2, hipStream_t s);       // one idle wavefront for that long; its start / end ticks to stamp2 (lane-overlap probe)
hipError_t launch_count_nonfinite(const void* p, long long n, int kind, unsigned* out, hipStream_t s);      // debug: kind 0 fp32, 1 bf16, 2 f16, 3 fp64

// ---- layout / elementwise / codec kernels (dd_misc.hip) ----------------------------------------
// dst layout: plain NHWC when blocked == 0 (naive path), else the activation layout of dd_elem.h
hipError_t launch_nchw_to_nhwc(const float* src, void* dst, int ek, int B, int C, int h, int w, int blocked, hipStream_t s);
// same, into a destination with Cd >= C channels (zero-filled beyond C; Cd a multiple of 16): pyramid widths that are not a
// multiple of the 32-channel activation block (MPViT-small's 216 -> 224)
hipError_t launch_nchw_to_nhwc_padded(const float* src, void* dst, int ek, int B, int C, int Cd, int h, int w, int blocked, hipStream_t s);
// Swin variant: bilinear (align_corners=True) upsample of the NCHW fp32 condition map (B,C,ch,cw) to (h,w), written in the
// channel-blocked activation layout (reference ...swin_addHAHI.py:332: F.interpolate(..., mode='bilinear', align_corners=True))
hipError_t launch_upsample_to_blocked(const float* src, void* dst, int ek, int B, int C, int ch, int cw, int h, int w, hipStream_t s);
// channel-blocked activations -> NCHW fp32 (tiled through LDS), and adaptive_avg_pool2d on channel-blocked activations
hipError_t launch_blocked_to_nchw(const void* src, int ek, float* dst, int B, int C, int h, int w, int accumulate, hipStream_t s);
hipError_t launch_adaptive_pool_blocked(const void* src, void* dst, int ek, int B, int C, int ih, int iw, int oh, int ow, hipStream_t s);
// bilinear upsample (align_corners=True) between two channel-blocked tensors (Swin condition map kept in the handle)
hipError_t launch_upsample_blocked(const void* src, void* dst, int ek, int B, int C, int ch, int cw, int h, int w, hipStream_t s);
hipError_t launch_nhwc_to_nchw_f32(const void* src, int ek, float* dst, int B, int C, int h, int w, int blocked, hipStream_t s);
// out(NCHW) = c1*x + c2*relu(gn4(y4))  (mode 0, final DDIM update) or relu(gn4(y4)) (mode 1, eps)
hipError_t launch_final(const float* x, const float* y4, const double* stats, const float* gamma, const float* beta,
                        const float* c1c2, int step, int mode, float* out_nchw, int B, int h, int w, hipStream_t s);
// etab[t][tap][co] = sum_c w3[co][c][tap] * emb[t][c] (tap < 9), etab[t][9][co] = sum over taps; w3 = pred.0 weight OIHW (64,256,3,3)
hipError_t launch_etab(const float* w3_oihw, const float* emb, float* etab, hipStream_t s);
// Swin hoist: ttab[k][..][64] for the T loop steps (timesteps ts[k]) of an h x w image from the fp32 OIHW weights of convA, convB, pred.0 and the
// embedding table; scratch: T * R_h * R_w * 576 floats (R = min(n, 7))
hipError_t launch_swin_ttab(const float* wa_oihw, const float* wb_oihw, const float* w3_oihw, const float* emb, const long long* ts, int T,
                            int h, int w, float* scratch, float* ttab, hipStream_t s);
// Swin hoist, 5x5 form: w5 (64,256,5,5) OIHW fp32 and the tap-pair products pairp[e][d][ci][co] = sum_cm W3[co][cm][e] * WB[cm][ci][d] (81 x 256 x 64)
// ... and kside: the four 5-tap line kernels of the border correction in MFMA fragment order, bf16, f16, then fp32 (SWIN_KSIDE_BYTES; swin_bcorr_line_kernel)
constexpr size_t SWIN_KSIDE_BYTES = (size_t)4 * 5 * COND_C * HID_C * (2 + 2 + 4);
hipError_t launch_swin_compose(const float* wb_oihw, const float* w3_oihw, float* w5_oihw, float* pairp, void* kside, hipStream_t s);
// bcorr[b][ring(q)][co] = sum over taps e with q + e outside the image, taps d with q + e + d inside, ci: pairp[e][d][ci][co] * sa[b][q + e + d][ci]
// (sa: convA's result in the activation layout, element kind ek)
hipError_t launch_swin_bcorr(const void* sa, int ek, const float* pairp, const void* kside, float* bcorr, int B, int h, int w, hipStream_t s);
hipError_t launch_add_noise(const float* x0, const float* noise, const long long* t, const float* acp, int n_train,
                            float* out, int B, long long per_sample, hipStream_t s);
struct CodecWeights {   // device pointers, BatchNorm already folded (eval mode)
  const float* enc_w0;  // [16][9]      conv 1->16 s2, scale folded
  const float* enc_b0;  // [16]
  const float* enc_w1;  // [16][16][9]  conv 16->16, scale folded (OIHW)
  const float* enc_b1;  // [16]
  const float* enc_w1t; // [9][16 in][16 out] the same weights tap-major (enc1_kernel: a (tap, ci) row of couts is one scalar load)
  const float* dec_w0;  // [16 in][16 out][4][4] convT, scale folded over out
  const float* dec_b0;  // [16]
  const float* dec_w0t; // [4][4][16 in][16 out] the same weights tap-major (fused decoder kernel)
  const float* dec_w1;  // [16][9]      conv 16->1
  float dec_b1;
};
hipError_t launch_encode(const CodecWeights& cw, const float* depth, float* tmp_nhwc, float* latent_nchw,
                         int B, int H, int W, hipStream_t s);
hipError_t launch_decode(const CodecWeights& cw, const float* latent_nchw, float* tmp_nhwc, float* depth,
                         int B, int h, int w, hipStream_t s);

// ---- backward pieces (dd_bwd.hip): SURVEY.md 8f rank 2 -------------------------------------------------
// An activation tensor as the kernels above store it: C channels, HW pixels per sample; blocked = channel-blocked layout
// of dd_elem.h (C >= 32), otherwise plain NHWC.
struct ActView { const void* p; int ek; int blocked; int C; long long HW; };
hipError_t launch_gn_bwd_reduce(const ActView& ga, const ActView& y, const double* stats, const float* gamma, const float* beta,
                                double* out_bc2, int B, hipStream_t s);
hipError_t launch_gn_bwd_apply(const ActView& ga, const ActView& y, const double* stats, const float* gamma, const float* beta,
                               const double* dgb, const ActView& gy, const ActView& act, const ActView& cond, const float* emb,
                               const long long* tvec, int t_base, int t_bstride, int B, hipStream_t s);
// vectorised forms for channel-blocked bf16 / f16 tensors with C = 64 / 256; sums are [B][C][4] doubles (see dd_bwd.hip).
// ek = kind of the gradients / materialised activations, yk = kind of the stored conv output y and of the condition map
// (yk == ek, or ek bf16 with yk f16: the mode EK_BF16M)
hipError_t launch_gn_bwd_reduce_blocked(const void* ga, const void* y, int ek, int yk, const double* stats, const float* gamma, const float* beta,
                                        double* out_bc4, int B, int C, long long HW, hipStream_t s);
hipError_t launch_gn_bwd_apply_blocked(const void* ga, const void* y, int ek, int yk, const double* stats, const float* gamma, const float* beta,
                                       const double* sums_bc4, void* gy, void* act, const void* cond, const float* emb,
                                       const long long* tvec, int t_base, int t_bstride, int B, int C, long long HW, hipStream_t s);
hipError_t launch_gn_param_grad4(const double* sums_bc4, const double* stats, const float* gamma, float* dgamma, float* dbeta, float* dbias,
                                 float* demb, const long long* tvec, int t_base, int t_bstride, int B, int C, long long HW, hipStream_t s);
hipError_t launch_channel_sum_blocked(const void* v, int ek, float* out, int B, int C, long long HW, hipStream_t s);
// adjoint of the align_corners bilinear upsample: blocked 2-byte (h, w) gradient -> NCHW fp32 (ch, cw) gradient
hipError_t launch_upsample_adjoint(const void* g, int ek, float* dst, int B, int C, int ch, int cw, int h, int w, int accumulate, hipStream_t s, bool tiled = true);
hipError_t launch_channel_sum(const ActView& v, float* out, const long long* rows, int t_base, int t_bstride, int B, hipStream_t s);
hipError_t launch_add_inplace(float* dst, const float* src, long long n, hipStream_t s);
hipError_t launch_gn_param_grad(const double* dgb, float* dgamma, float* dbeta, int B, int C, hipStream_t s);
hipError_t launch_naive_wgrad(const ActView& gy, const ActView& a, float* dw_oihw, int B, int h, int w, hipStream_t s);
// MFMA weight gradient (dd_wgrad.hip), bf16 / f16 operands in the activation layouts, fp32 atomics into dw [CO][CI][3][3]
size_t wgrad_workspace_bytes(int CO, int CI, int B, int h, int w);     // per-slab partial sums
hipError_t launch_wgrad_mfma(const void* gy, const void* a, float* dw_oihw, float* workspace, int ek, int CO, int CI, int B, int h, int w, hipStream_t s,
                             int share = 1);      // share = concurrent callers splitting the GPU (lanes): fewer partial-sum slabs each
// chain rule of x_{k+1} = c1_k x_k + c2_k eps: mode 0: ga = c2_k * g;  mode 1: g = c1_k * g + ga   (c1c2 = [T][2] device table)
hipError_t launch_bwd_chain(float* g, float* ga, const float* c1c2, int k, int mode, long long n, hipStream_t s);
hipError_t launch_view_copy(const ActView& src, const ActView& dst, int B, hipStream_t s);
hipError_t launch_view_to_nchw(const ActView& v, float* dst, int B, int accumulate, hipStream_t s);

// ---- naive cross-check path (dd_naive.hip) -------------------------------------------------------
hipError_t launch_naive_conv3x3(const float* in_nhwc, const float* w_oihw, const float* bias, float* out_nhwc,
                                int B, int h, int w, int cin, int cout, hipStream_t s);
hipError_t launch_naive_gn_stats(const float* y_nhwc, double* stats, int B, int h, int w, int C, hipStream_t s);
// out = relu(gn(y)) [+ cond + emb[t]]
hipError_t launch_naive_gn_apply(const float* y, const double* stats, const float* gamma, const float* beta,
                                 const float* cond, const float* emb, const long long* tvec, int t_base, int t_bstride,
                                 float* out, int B, int h, int w, int C, hipStream_t s);
// x_out = c1*x + c2*eps (eps already normalised)
hipError_t launch_naive_axpby(const float* x, const float* eps, const float* c1c2, int step, float* out,
                              long long n, hipStream_t s);

}  // namespace dd
