// dd_igemm2_cfg.h -- per-layer tiling constants of the fused convolution kernels (dd_igemm2.hip).
// Variants that were measured slower on MI355X and removed in round 2 (profiles/history/r01_run30_power_and_variants.md,
// profiles/history/r02_run1_ddimloss_and_winograd.md): wave-specialised staging waves, a 16x32 / 8-wave / 9-taps-per-stage conv3 tile with
// and without ping-pong halves, persistent conv1 / conv4 workgroups, in-loop interleaving of the prologue, Winograd F(2x2,3x3).
// Round 3 (profiles/history/r03_run2_variants.md): the next stage's weight DMA issued between the tap groups of the MFMA block, conv1 at eight
// waves per SIMD, eight-wave "latency variants" of conv2 / conv3, a persistent conv3.
#pragma once
#include "dd_elem.h"

// ConvParams::ablate (timing experiments: skip parts of the kernel) is honoured only in builds with -DDD_ABLATE=1;
// the default build folds every check away so the main loop is straight-line code.
#ifndef DD_FRAG_DEPTH
#define DD_FRAG_DEPTH 2
#endif
#ifndef DD_RAW_DEPTH
#define DD_RAW_DEPTH 2
#endif
#ifndef DD_ABLATE
#define DD_ABLATE 0
#endif
// tiling of the conv3-shaped layers (256 -> 64: layers 3, 7, 8, 9) in the 2-byte kinds:
//   0 = 8x32 pixels, 32-channel chunks, 3 taps per stage (24 MFMAs per wave between barriers)
//   1 = 8x32 pixels, 16-channel chunks, 9 taps per stage (36 MFMAs per wave between barriers)
//   2 = 16x32 pixels, 4 waves x (128 pixels x 64 couts), 16-channel chunks, 9 taps per stage (72 MFMAs between barriers, 0.75 LDS reads per MFMA)
// Round 3: tiling 2 is ALSO instantiated beside the default as the kernel ids BIG_CONV3C / BIG_CONV3H (layers 8 / 9) and picked per launch when
// the tiles do not fit the chip at once -- under the two concurrent lanes it is worth +1.8 % at B=4 and +4 % at B=8 (profiles/history/r03_run2_variants.md).
#ifndef DD_C3
#define DD_C3 1
#endif
// conv2 (64 -> 256, two 128-cout splits): 1 = ONE workgroup per tile walks both cout splits over the same staged input patch (one
// prologue -- partial sums, table, patch fetch, normalisation -- per tile instead of two; the second split's weight stream follows
// the first's without a gap); 0 = one workgroup per (tile, split)
#ifndef DD_C3_2_RD1
#define DD_C3_2_RD1 1      // 16x32 conv3 tiles: raw patch one chunk ahead only (registers: two workgroups per CU need <= 128 VGPRs beside the accumulators)
#endif
// Swin fuse convs (256 -> 256, layers 5 / 6) in the 2-byte kinds: 0 = 32-channel chunks, one tap per stage (16 MFMAs per wave between
// barriers); 1 [default since round 2: convB -4.6 %, convA -5 % with the two options below] = 16-channel chunks, three taps per stage
// (24 MFMAs between barriers, 46 KB of LDS)
#ifndef DD_SWIN_TG3
#define DD_SWIN_TG3 1
#endif
// s_setprio level of a wave while it is inside a stage's MFMA block (0 = off)
#ifndef DD_SETPRIO
#define DD_SETPRIO 0
#endif
#ifndef DD_SWIN_FD2
#define DD_SWIN_FD2 1      // convA with DD_SWIN_TG3: two-deep fragment registers (the 32-channel-chunk prologue left no room for them)
#endif
#ifndef DD_SWIN_RD2
#define DD_SWIN_RD2 1      // convB with DD_SWIN_TG3: raw patch two chunks ahead
#endif
#ifndef DD_CONV2_DUAL
#define DD_CONV2_DUAL 1
#endif
// 1 = kernels take the timestep from ConvParams::t_known when the host knew it (0: always through the tvec load; A/B switch)
#ifndef DD_T_KNOWN
#define DD_T_KNOWN 1
#endif
// hoisted condition term conv3(cond) (layer 8 -> layer 9) in the 2-byte modes: 1 = stored as f16 quads (8 bytes per lane and register quad:
// 128 instead of 256 bytes per pixel and step in conv3's prologue; it is added to accumulators whose result is stored as f16 anyway),
// 0 = fp32.  The fp32 and split-f16 modes always keep fp32.
#ifndef DD_CADD_F16
#define DD_CADD_F16 1      // measured (round 3, one box): conv3 155 -> 138 us at KITTI B=4, 46.5 -> 44.9 at B=1; loop 7.30 -> 7.03 ms
#endif
// conv4 (64 -> 16) in the 2-byte kinds: 1 = 16-channel chunks, nine taps per stage (four chunks, double-buffered 11-KB patches, 40 KB of LDS:
// three to four workgroups per CU instead of two, the next chunk's loads fly under the current chunk's MFMAs); 0 = one 64-channel patch
#ifndef DD_C4_CK16
#define DD_C4_CK16 1       // measured (round 3): conv4 36.7 -> 34.3 us at KITTI B=4, neutral at B=1
#endif
// per-workgroup phase timestamps (ConvParams::prof); compiled in by tools/phase_prof.py only
#ifndef DD_PHASE_PROF
#define DD_PHASE_PROF 0
#endif

namespace dd {

template <int EKM_, int LAYER_ID_> struct Cfg2 {
  // kernel ids BIG_CONV3C / BIG_CONV3H (dd_kernels.h) = layers 8 / 9 on 16x32-pixel tiles
  static constexpr bool BIG = LAYER_ID_ == BIG_CONV3C || LAYER_ID_ == BIG_CONV3H || LAYER_ID_ == SWIN_PRED5B_H;
  // kernel ids SWIN_CONVA_H / SWIN_PRED_H (dd_kernels.h) = layers 5 / 7 of the Swin denoiser with the step-invariant terms hoisted
  // SWIN_PRED5_H = pred.0 o convB as one 5x5 convolution (layer 7's tiling with five taps per stage)
  static constexpr bool PRED5 = LAYER_ID_ == SWIN_PRED5_H || LAYER_ID_ == SWIN_PRED5B_H;
  static constexpr bool HOIST_A = LAYER_ID_ == SWIN_CONVA_H, ADD_T = LAYER_ID_ == SWIN_PRED_H || PRED5;
  static constexpr bool IN_NCHW = LAYER_ID_ == CONV3C_NCHW;     // layer 8 on the caller's NCHW fp32 tensor (dd_kernels.h)
  static constexpr int LAYER_ = (LAYER_ID_ == BIG_CONV3C || IN_NCHW) ? 8 : (LAYER_ID_ == BIG_CONV3H || LAYER_ID_ == ONE_CONV3H) ? 9 : HOIST_A ? 5 : ADD_T ? 7 : LAYER_ID_;
  // EKM_ = element kind or the mode EK_BF16M (dd_kernels.h).  In that mode only the layers that CHANGE kind between storage and operands
  // are instantiated here -- conv2 / conv3 / hoisted conv3 / Swin convA (f16 in, bf16 operands), the producers of f16 tensors in front
  // of them (conv2, conv3, Swin pred.0, the level-0 lateral conv of the condition FPN); the launcher sends every other layer to its
  // plain f16 (conv1, conv4, once-per-image conv3(cond)) or bf16 (Swin convB, inner FPN layers, data gradients) instantiation.
  static constexpr bool MX = EKM_ == EK_BF16M;
  // EKM_ == EK_F16S (split f16, DD_PREC_F16X3; dd_kernels.h): tensors in HBM are fp32 (the fp32 mode's layouts), the LDS patch and the
  // packed weights carry TWO f16 planes (hi, lo) and every (weight, pixel) fragment pair costs three MFMAs.  Instantiated for the
  // denoiser's layers 1..9 and (round 4) the once-per-image layers in front of them -- the condition FPN 10..18 / 24..26 and the HAHI neck
  // 30..41 / 54..65, which the split / refined f16 modes used to run on the fp32-operand kernels (a fifth of this rate); one tiling for all
  // of them: 16-channel chunks, 3 taps per stage (conv1: 9, the 1x1 layers: 1), 64-cout workgroup tiles.
  // EKM_ == EK_F16R (refined f16, DD_PREC_F16R; dd_kernels.h): instantiated for the layers whose kernel differs from the f16 mode's only --
  // conv1 (its weights as an f16 pair against a single-plane patch, y1 stored f16) and the producer of y3 -- the hoisted conv3 of the Res denoiser
  // (layer 9 and its tile forms) / the 5x5 form pred.0 o convB of the Swin denoiser: f16 operands and input, the hoisted term read as block-scaled
  // int16, y3 written as int16 with a per-pixel scale; every other layer of the mode runs its EK_F16 (conv2, Swin convA', conv4 via dd_thin.hip)
  // or EK_F16S (the once-per-image chain) form.
  static constexpr bool RF = EKM_ == EK_F16R;
  static_assert(!RF || LAYER_ == 1 || LAYER_ == 9 || PRED5, "EK_F16R is instantiated for conv1 and the hoisted conv3 / Swin 5x5 forms only");
  static constexpr bool SPLIT = EKM_ == EK_F16S || (RF && LAYER_ == 1);
  static_assert(!IN_NCHW || EKM_ == EK_F16S, "the NCHW-reading layer 8 is a split-f16 kernel");
  static constexpr bool WONLY = RF && LAYER_ == 1;             // split weights against a single-plane patch: W.P = Whi.P + Wlo.P
  static constexpr int NPL = SPLIT ? 2 : 1;                    // operand planes of a packed weight stage
  static constexpr int NPLP = (SPLIT && !WONLY) ? 2 : 1;       // operand planes of the LDS patch
  static constexpr bool Q15 = RF && (LAYER_ == 9 || PRED5);    // hoisted term read as scaled int16 quads, y3 written as int16 with a per-pixel scale
  static constexpr int EK = MX ? (int)EK_BF16 : (SPLIT || RF) ? (int)EK_F16 : EKM_;          // MFMA operand kind = kind of the LDS patch and of the packed weights
  static constexpr int LAYER = LAYER_;
  static constexpr int IN_K = SPLIT ? (int)EK_F32 : (MX && (LAYER_ == 2 || LAYER_ == 3 || LAYER_ == 5 || LAYER_ == 9)) ? (int)EK_F16 : EK;      // stored input (and condition map)
  static constexpr int OUT_K = RF ? (int)EK_F16 : SPLIT ? (int)EK_F32 : (MX && (LAYER_ == 2 || LAYER_ == 3 || LAYER_ == 7 || LAYER_ == 9 || LAYER_ == 10 || LAYER_ == 15 || LAYER_ == 24)) ? (int)EK_F16 : EK;
  static_assert(!MX || IN_K != EK || OUT_K != EK, "EK_BF16M is instantiated only for the layers that change kind");
  static_assert(!SPLIT || (LAYER_ >= 1 && LAYER_ <= 18) || (LAYER_ >= 24 && LAYER_ <= 26) || (LAYER_ >= 30 && LAYER_ <= 41) || (LAYER_ >= 54 && LAYER_ <= 65),
                "EK_F16S is instantiated for the forward layers only (denoiser, condition FPN, HAHI neck)");
  static_assert(!SPLIT || LAYER_ID_ != SWIN_PRED_H, "split f16: the hoisted Swin plans always run the 5x5 form");
  static constexpr int ESZ = ElemSize<EK>::V;
  // layers 1..4: conv1..conv4 of the Res denoiser.  Swin/MPViT variant (reference ...swin_addHAHI.py:321-382):
  //   5 = upsample_fuse.convA 256->256 (prologue relu(gn2(y2)) + up(cond) + E[t]), 6 = upsample_fuse.convB 256->256
  //   (raw input, no norm / activation in between: ConvModule(norm_cfg=None, act_cfg=None)), 7 = pred.0 256->64 on a raw input
  // Res denoiser with the condition term hoisted (optional): 8 = conv3 applied ONCE per image to the raw condition map (fp32
  //   out, no bias / statistics), 9 = conv3 on relu(gn2(y2)) only, epilogue adds layer 8's output and the E[t] tap sums
  // Condition aggregation (FPN of the Res head, reference ...res.py:56-84,108-118; eval-mode BN folded into weights / bias):
  //   10..13 = conv_lateral[0..3]: Conv3x3 (64|128|256|512 -> 256) + BN + ReLU, then "+ top-down term" (optional addend)
  //   14     = conv_up[j]: ConvTranspose2d(256->256, k2, s2) + BN + ReLU written as a 1x1 conv with 4 x 256 output
  //            "channels" (one block per output parity (dy,dx)) whose epilogue scatters to pixel (2y+dy, 2x+dx)
  //   15..18 = the same lateral convs for the Swin-L pyramid (192|384|768|1536 -> 256; reference ...res_swin_add.py:31,57-84)
  // Backward: 20..23 = data gradients of conv4, conv3, conv2, conv1 -- the same implicit GEMM with
  //   W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx] on the raw GroupNorm-backward result: 16->64, 64->256, 256->64, 64->16
  //   24..26 = lateral convs of the MPViT-small pyramid (reference ...res_mpvit_HAHI.py:32: 128 | 216 | 288 | 288 -> 256; 216 is carried
  //            as 224 = 7 blocks of 32 with zero channels / zero weights; levels 2 and 3 share layer 26)
  // HAHI neck of the Swin-L heads with the attention off (reference src/model/necks/hahi.py:60-97,170-173,196-197,226-272; eval-mode BN
  // folded): per pyramid level i (C_i = 192 << i, embedding 512)
  //   30 + i = lateral_convs[i]: 1x1 C_i -> C_i        34 + i = conv_proj / trans_proj[i-1]: 1x1 C_i -> 512
  //   38 + i = conv_fusion / trans_fusion[i-1]: 3x3 (C_i + 512) -> C_i on the channel concatenation
  // all + ReLU, raw inputs.  The concatenation is free in the channel-blocked layout: the two 1x1 convs write their couts at a channel
  // offset of ONE buffer (ConvParams::out_coff / out_cstride), the projection reads the lateral result from it (in_coff / in_cstride).
  //   54..65 = the same twelve convolutions for the MPViT-small pyramid (reference ...res_mpvit_HAHI.py:32,51-53: 128 | 216 | 288 | 288,
  //   embedding 512): 216 is carried as 224 channels (zero channels / zero weights, as in dd_condition), couts round up to whole 64-cout
  //   workgroup tiles (224 -> 256, 288 -> 320: the padding couts are computed on zero weights and never stored)
  static constexpr bool NECK_MP = (LAYER >= 54 && LAYER <= 65);
  static constexpr bool IS_NECK = (LAYER >= 30 && LAYER <= 41) || NECK_MP;
  static constexpr int NECK_BASE = NECK_MP ? 54 : 30;
  static constexpr int NECK_LVL = IS_NECK ? (LAYER - NECK_BASE) % 4 : 0;
  static constexpr int NECK_KIND = IS_NECK ? (LAYER - NECK_BASE) / 4 : -1;      // 0 lateral, 1 projection, 2 fusion
  static constexpr int NECK_C = NECK_MP ? (NECK_LVL == 0 ? 128 : NECK_LVL == 1 ? 224 : 288) : (192 << NECK_LVL);
  static constexpr bool IS_LAT = (LAYER >= 10 && LAYER <= 13) || (LAYER >= 15 && LAYER <= 18) || (LAYER >= 24 && LAYER <= 26);
  static constexpr bool IS_DGRAD = (LAYER >= 20 && LAYER <= 23);
  static constexpr bool IS_UP = (LAYER == 14);
  static constexpr int KS = PRED5 ? 5 : (IS_UP || (IS_NECK && NECK_KIND != 2)) ? 1 : 3;      // kernel size
  static constexpr int HALO = KS / 2;
  static constexpr int NTAPS = KS * KS;
  static constexpr int CIN = IS_NECK ? (NECK_KIND == 2 ? NECK_C + 512 : NECK_C) : (LAYER == 1 || LAYER == 20) ? LATENT_C : (LAYER == 2 || LAYER == 4 || LAYER == 21 || LAYER == 23) ? HID_C : IS_LAT ? (LAYER >= 24 ? (LAYER == 24 ? 128 : LAYER == 25 ? 224 : 288) : LAYER >= 15 ? (192 << ((LAYER - 15) & 3)) : (64 << ((LAYER - 10) & 3))) : COND_C;
  static constexpr int COUT = IS_NECK ? (NECK_KIND == 1 ? 512 : NECK_C) : IS_UP ? 4 * COND_C : IS_LAT ? COND_C
                            : (LAYER == 21) ? COND_C : (LAYER == 23) ? LATENT_C
                            : (LAYER == 1 || LAYER == 3 || LAYER >= 7) ? HID_C : (LAYER == 4) ? LATENT_C : COND_C;
  static constexpr bool RELU_OUT = IS_LAT || IS_UP || IS_NECK;   // epilogue: relu(acc + bias)
  static constexpr bool SCATTER = IS_UP;                         // epilogue: cout block -> output parity of a 2x upsampled tensor
  static constexpr int COUT_PAD = (COUT < 32) ? 32 : IS_NECK ? ((COUT + 63) / 64) * 64 : COUT;
  static constexpr bool C3SHAPE = (LAYER == 3 || LAYER == 7 || LAYER == 8 || LAYER == 9);
  static constexpr int C3 = (C3SHAPE && ESZ == 2 && !SPLIT) ? (BIG ? 2 : DD_C3) : 0;
  static_assert(!BIG || (ESZ == 2 && !SPLIT), "big-tile forms: 2-byte kinds only");
  static constexpr bool SWIN3 = DD_SWIN_TG3 && (LAYER == 5 || LAYER == 6) && ESZ == 2 && !SPLIT;
  static constexpr bool C4K16 = DD_C4_CK16 && LAYER == 4 && ESZ == 2 && !SPLIT;
  static constexpr int CK = (SPLIT || C4K16) ? 16 : (LAYER == 1 || LAYER == 20) ? 16 : (LAYER == 2 || LAYER == 4 || LAYER == 21 || LAYER == 23) ? (128 / ESZ) : (C3 != 0 || SWIN3) ? 16 : (64 / ESZ);
  static constexpr int TG = PRED5 ? 5 : SPLIT ? (LAYER == 1 ? 9 : KS == 1 ? 1 : 3) : (LAYER == 1 || LAYER == 20 || C3 != 0 || C4K16) ? 9 : (LAYER == 22 || LAYER == 23) ? 3
                          : SWIN3 ? 3 : (LAYER == 2 || LAYER == 5 || LAYER == 6 || LAYER >= 10) ? 1 : 3;
  static constexpr int NT = (IS_NECK || (SPLIT && COUT >= 64)) ? 64 : (COUT >= COND_C) ? 128 : COUT_PAD;
  static constexpr int SPW = (LAYER == 2 && ESZ == 2 && DD_CONV2_DUAL && !SPLIT) ? 2 : 1;      // cout splits one workgroup walks (over one staged patch)
  static constexpr bool STATS = !(LAYER == 5 || LAYER == 6 || LAYER == 8 || LAYER >= 10);   // a GroupNorm follows this convolution
  static constexpr bool ADD_C = (LAYER == 9);                  // epilogue adds the hoisted condition / embedding terms
  static constexpr bool ADD_ACC = ADD_C || ADD_T;              // accumulators start at the hoisted per-image term (ConvParams::cadd)
  static constexpr bool CADD16 = DD_CADD_F16 && ESZ == 2 && !SPLIT && !RF && (LAYER == 8 || LAYER == 9 || ADD_T);   // the hoisted term travels as f16
  // conv1 / conv4 are latency-bound (18 MFMAs per 32-pixel block): 8 waves of one block each shorten every wave's
  // dependent chain (measured: 4x32 tiles with 4 waves were no faster for conv1 and slower for conv4 - more halo and
  // weight traffic); conv2 / conv3 and the Swin convs keep 4 waves x 2 blocks (fewer LDS reads per MFMA)
  static constexpr int TH = (C3 == 2) ? 16 : 8, TW = 32;
  static constexpr int WAVES = (LAYER == 1 || LAYER == 4 || LAYER == 20 || LAYER == 23) ? 8 : 4;
  static constexpr int THREADS = WAVES * 64;
  static constexpr int WM = (TH * TW) / (32 * WAVES);
  static constexpr int WN = NT / 32;
  static constexpr int PRO = HOIST_A ? PRO_GN : (LAYER == 1) ? PRO_X : (LAYER == 3 || LAYER == 5) ? PRO_GN_ADD : (LAYER == 9) ? PRO_GN : (LAYER >= 6) ? PRO_RAW : PRO_GN;
  static constexpr int IN_ESZ = (LAYER == 1 || SPLIT) ? 4 : ESZ;
  static constexpr int OUT_ESZ = RF ? 2 : (LAYER == 4 || LAYER == 8 || LAYER == 23 || SPLIT) ? 4 : ESZ;
  static constexpr int PH = TH + 2 * HALO, PW = TW + 2 * HALO;
  static constexpr int ROWB = CK * ESZ;                  // 32 / 64 / 128 bytes
  static constexpr int PPP = ROWB / 16;
  static constexpr int LOG2_PPP = (PPP == 2) ? 1 : (PPP == 4) ? 2 : 3;
  static constexpr int RPB = 256 / ROWB;                 // rows per 256-B LDS bank row
  static constexpr int EPP = 16 / ESZ;
  static constexpr int NKQ = PPP / 2;                    // MFMA k-steps per tap (16 B from each half-wave)
  static constexpr int NCHUNK = CIN / CK;
  static constexpr int NTG = NTAPS / TG;
  static constexpr int NSTAGE = NCHUNK * NTG;
  // kernel id ONE_CONV3H: the hoisted conv3 on 8x32 tiles with ONE patch buffer -- the next chunk's patch is written into the buffer the MFMAs just
  // read, behind a second workgroup barrier per stage, instead of into a twin buffer under the running stage.  63 -> 52 KB of LDS = THREE
  // workgroups per CU (12 waves, three per SIMD; the kernel needs 160 VGPRs) where the two-buffer form has two: a wave spends ~20 % of a stage in
  // its MFMA block and the rest waiting for memory, the weight DMA or a barrier -- it lacks co-resident waves, not issue slots.  Measured
  // (round 4, profiles/history/r04_call3_*): conv3 at KITTI B=4 on one stream 140 -> 132 us (bf16), 147 -> 137 (f16r); neutral under two lanes and at B=1
  static constexpr bool ONEBUF = LAYER_ID_ == ONE_CONV3H;
  static_assert(!ONEBUF || (C3 == 1 && NCHUNK > 1), "one-buffer form: the 8x32 tiling of the 2-byte kinds");
  static constexpr int NPB = (NCHUNK > 1 && !ONEBUF) ? 2 : 1;       // patch buffers
  static constexpr int PATCH_PLANE = PH * PW * ROWB;      // one operand plane of a patch buffer (split f16: hi plane, then lo plane)
  static constexpr int PATCH_BYTES = NPLP * PATCH_PLANE;
  static constexpr int W_PLANE = TG * NT * ROWB;          // one operand plane of a weight stage
  static constexpr int W_BYTES = NPL * W_PLANE;
  static constexpr int NWB = (NSTAGE > 1) ? 2 : 1;       // weight ring slots
  static constexpr int W_OFF = NPB * PATCH_BYTES;        // LDS byte offset of the weight ring
  static constexpr int CTAB = (PRO == PRO_RAW) ? 0 : (LAYER == 1) ? LATENT_C : CIN;   // channels of the prologue GroupNorm table
  static constexpr int SCR_FLOATS = 2 * 8 * 8;                               // cross-wave scratch of the statistics epilogue: 8 doubles per wave
  static constexpr int NTABS = (PRO == PRO_GN_ADD) ? 3 : 2;                   // a, b (and the embedding row e) of the prologue GroupNorm
  static constexpr int TAB_FLOATS = NTABS * CTAB + NT * SPW + (ADD_C ? 10 * HID_C : 0) + SCR_FLOATS;   // GroupNorm table, bias, E[t] tap sums, scratch
  static constexpr int SMEM_BYTES = NPB * PATCH_BYTES + NWB * W_BYTES + TAB_FLOATS * 4;
  static constexpr int ITEMS = PH * PW * PPP;
  static constexpr int NIT = (ITEMS + THREADS - 1) / THREADS;      // staging items per thread
  static constexpr int NLD = EPP * IN_ESZ / 16;
  static constexpr int PIXSTRIDE = (CIN >= ACT_CB) ? ACT_CB : CIN; // elements between pixels of one channel block
  // two-deep fragment registers in the MFMA loop (next group's ds_reads issued between this group's MFMAs); the Swin convA
  // prologue (GroupNorm + upsampled condition + embedding on 256 channels, 128 couts per wave) has no registers to spare
  // raw-patch register slots: layers whose prologue reads ONE tensor (no aux term) and has several channel chunks fetch two
  // chunks ahead -- measured on MI355X the global-load latency under load (4-5 us) exceeds one chunk of MFMA work
  static constexpr int RAW_DEPTH = (DD_RAW_DEPTH == 2 && !SPLIT && (LAYER == 9 || LAYER == 7 || LAYER == 22 || ((LAYER == 6 || HOIST_A) && SWIN3 && DD_SWIN_RD2)) && !(C3 == 2 && DD_C3_2_RD1)) ? 2 : 1;
  static constexpr int FRAG_DEPTH = (LAYER == 5 && !SPLIT && !(SWIN3 && DD_SWIN_FD2)) ? 1 : DD_FRAG_DEPTH;
  // registers: two workgroups per CU for conv2 / conv3 (8 waves = 2 per SIMD, <= 256 VGPR+AGPR);
  // conv1 / conv4 want >= 2 eight-wave workgroups per CU (<= 128 registers)
  static constexpr int MIN_WAVES_PER_SIMD = (WAVES == 8) ? ((SMEM_BYTES <= 80 * 1024) ? 4 : 2) : (ONEBUF && SMEM_BYTES <= 160 * 1024 / 3) ? 3 : ((SMEM_BYTES <= 80 * 1024) ? 2 : 1);   // 80 KiB = half the CU's LDS
  static_assert(CIN % CK == 0 && NTAPS % TG == 0 && COUT_PAD % NT == 0, "tiling");
  static_assert(PATCH_BYTES % 16 == 0 && W_BYTES % 1024 == 0, "LDS carve / DMA granularity");
  static_assert(WAVES <= 8 && (COUT_PAD / NT) % SPW == 0 && (SPW == 1 || NCHUNK == 1), "scratch size; splits per workgroup");
  static_assert(ROWB == 32 || ROWB == 64 || ROWB == 128, "swizzle derivation");
  static_assert(TW == 32, "one 32-pixel MFMA block == one tile row (column-only swizzle, lane == column)");
  static_assert(THREADS % PPP == 0 && NIT <= 32, "per-thread piece index is constant; masks fit 32 bits");
  static_assert(TG == 1 || TG == 3 || TG == 9 || (TG == 5 && KS == 5), "tap decomposition");
};

// LDS swizzle: 16-B piece j of a row is stored at j ^ swz(k), k = the row's COLUMN index in the patch
// (weights: the cout row index).  Any 16 lanes with distinct k mod 16 then hit 16 distinct 16-B slots of
// the 256-B bank row, whatever the (wave-uniform) patch row is, because PW*ROWB is a multiple of 128 B.
template <int RPB, int PPP> __device__ __forceinline__ int swz16(int k) { return ((k / RPB) & (PPP - 1)) << 4; }

}  // namespace dd
