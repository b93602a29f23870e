// dd_api_internal.h -- what the translation units of the C ABI (include/ddepth.h) share: the device-buffer / plan / handle types and the
// functions one unit defines and another calls.  Round 5 split the 2 500-line dd_api.cpp by concern, no behaviour change:
//   dd_api.cpp          handle lifetime, schedule, options, counters, dd_condition / dd_neck_condition, dd_denoise / _trace / _once (lanes), q_sample,
//                       latent codec, debug fetches
//   dd_api_weights.cpp  parameter intake (dd_set_weight / _device), packing into the kernels' weight images, dd_commit_weights
//   dd_api_plans.cpp    per-shape plans (buffers, schedule tables, hoisted terms), the launch sequence of one denoiser call / one loop body, graphs' inputs
//   dd_api_train.cpp    gradients: dd_denoise_once_backward, dd_denoise_backward (lanes), dd_zero_grad / dd_get_grad
#pragma once
#include "../../include/ddepth.h"
#include "dd_kernels.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <tuple>
#include <vector>

using namespace dd;

namespace ddapi {

extern thread_local std::string g_create_error;

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { release(); }
  void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
  hipError_t alloc(size_t n) {
    release();
    if (n == 0) n = 16;
    hipError_t e = hipMalloc(&p, n);
    if (e == hipSuccess) bytes = n;
    return e;
  }
  template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

inline uint16_t host_f32_to_bf16(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
inline uint16_t host_f32_to_f16(float f) {
  _Float16 hf = (_Float16)f;       // round-to-nearest-even
  uint16_t u;
  std::memcpy(&u, &hf, 2);
  return u;
}

constexpr int NUM_EK = 3;     // weight images every convolution has: fp32, bf16, f16 (index = element kind)
constexpr int NUM_WIMG = 5;   // ... the denoiser's forward convolutions have a fourth: the split-f16 image of the mode EK_F16S (index WIMG_SPLIT),
constexpr int WIMG_SPLIT = 3; // and conv4 a fifth: the f16 image with the weights' lo halves stacked into its padding cout rows (EK_F16R, index WIMG_STACK)
constexpr int WIMG_STACK = 4;
inline int wimg_kind(int slot) { return slot == WIMG_SPLIT ? (int)EK_F16S : slot == WIMG_STACK ? (int)EK_F16R : slot; }     // image slot -> kind handed to the packers / conv_pack_geom2
inline int wimg_slot(int kind) { return kind == EK_F16S ? WIMG_SPLIT : kind; }
inline bool wimg_has(int slot, int fwd_layer) { return slot != WIMG_STACK || fwd_layer == 4; }     // which convolution carries which image
// precision -> element kind / mode of the fused kernels.  DD_PREC_BF16 is the mode EK_BF16M (bf16 operands on the large convolutions, f16
// storage and thin layers: dd_kernels.h) unless the handle option "bf16_storage" = 1 selects all-bf16 tensors (A/B and error budget).
inline int ek_of_precision(int prec, bool bf16_pure) {
  switch (prec) {
    case DD_PREC_FP32: return EK_F32;
    case DD_PREC_BF16: return bf16_pure ? EK_BF16 : EK_BF16M;
    case DD_PREC_F16: return EK_F16;
    case DD_PREC_F16X3: return EK_F16S;
    case DD_PREC_F16R: return EK_F16R;
    default: return -1;
  }
}
inline size_t ek_size(int ek) { return (ek == EK_F32 || ek == EK_F16S) ? 4 : 2; }     // bytes per STORED element (EK_F16S stores fp32)
inline int thin_kind(int ek) { return ek == EK_BF16M ? (int)EK_F16 : ek; }      // conv1 / conv4 / once-per-image conv3(cond): kernels and weights
constexpr int DD_PREC_LAST = DD_PREC_F16R;

constexpr int FPN_LEVELS = 4;
constexpr int FPN_CIN_RES[FPN_LEVELS] = {64, 128, 256, 512};       // ResNet pyramid widths (reference ...res.py:31 in_channels)
constexpr int FPN_CIN_SWIN[FPN_LEVELS] = {192, 384, 768, 1536};   // Swin-L pyramid widths (reference ...res_swin_add.py:31)
// MPViT-small pyramid of DDIMDepthEstimate_MPVIT_ADDHAHI (reference src/model/head/ddim_depth_estimate_res_mpvit_HAHI.py:32); same
// UpSample_add denoiser as the Swin heads (DD_VARIANT_SWIN).  216 is not a multiple of the 32-channel activation block: level 1 is
// carried with 224 channels (8 zero channels, zero weights).  The pyramid is recognised from the lateral weights' sizes (dd_set_weight).
constexpr int FPN_CIN_MPVIT[FPN_LEVELS] = {128, 216, 288, 288};
constexpr int FPN_CIN_MPVIT_PAD[FPN_LEVELS] = {128, 224, 288, 288};
constexpr int FPN_LAYER_MPVIT[FPN_LEVELS] = {24, 25, 26, 26};
enum { PYR_DEFAULT = 0, PYR_MPVIT = 1 };
inline const int* fpn_cin(int variant, int pyr) {
  return pyr == PYR_MPVIT ? FPN_CIN_MPVIT : (variant == DD_VARIANT_SWIN ? FPN_CIN_SWIN : FPN_CIN_RES);
}
inline const int* fpn_cin_pad(int variant, int pyr) { return pyr == PYR_MPVIT ? FPN_CIN_MPVIT_PAD : fpn_cin(variant, pyr); }
inline int fpn_lat_layer(int variant, int pyr, int level) {      // kernel layer id
  return pyr == PYR_MPVIT ? FPN_LAYER_MPVIT[level] : (variant == DD_VARIANT_SWIN ? 15 : 10) + level;
}

struct ConvLayer {                  // one Conv3x3 + its following GroupNorm
  int cin = 0, cout = 0;
  DevBuf wpack2[NUM_WIMG];          // packed for the fused kernels (pre-swizzled for LDS-DMA)
  DevBuf bias;                      // [cout padded to 32]
  DevBuf w_oihw;                    // naive path
  DevBuf wpackT[NUM_EK];            // fused backward: W' packed for the dgrad layer (23 - conv index) of dd_igemm2.hip
  DevBuf wT_oihw;                   // naive backward: W'[ci][co][ky][kx] = W[co][ci][2-ky][2-kx] (dgrad as a forward conv)
  DevBuf gamma, beta;               // GroupNorm affine [cout]
};

struct PlanKey {
  int B, h, w, ch, cw, T, prec, hoist;
  int keep = 0;        // option "keep_trajectory" (what dd_denoise_backward needs): 1 = the loop leaves every state x_k in Plan::xstash;
                       // 2 = ... and every step's raw conv outputs y1..y4 (Swin: + convA / convB results) in per-step slots
  int lane = 0;        // dd_denoise with option "streams" > 1 runs a batch as concurrent sub-batches: one plan (buffers, graph) per lane
  int lanes = 1;       // ... and how many lanes the call runs as: the tile shape of the hoisted conv3 pair depends on it (plan_big_tiles)
  bool operator<(const PlanKey& o) const {
    return std::tie(B, h, w, ch, cw, T, prec, hoist, keep, lane, lanes) < std::tie(o.B, o.h, o.w, o.ch, o.cw, o.T, o.prec, o.hoist, o.keep, o.lane, o.lanes);
  }
};

struct Plan {
  PlanKey key{};
  int ek = EK_F32;
  DevBuf x[2];        // fp32 NHWC state ping-pong
  std::shared_ptr<DevBuf> cond;   // condition map at latent size (activation layout / element kind; fp32 NHWC for naive); shared by every
                                  // plan of one (B, h, w, element kind) so that dd_condition can write it in place
  const void* cond_alias = nullptr;   // lanes of a batch whose condition map dd_condition left in the whole batch's buffer: this lane's images inside it
  const void* graph_cond = nullptr;   // the condition pointer the captured graph holds
  void* cond_ptr() const { return cond_alias ? const_cast<void*>(cond_alias) : (cond ? cond->p : nullptr); }      // (null: a refined-f16 Res plan that has only ever read the caller's tensor in place)
  DevBuf y1, y2, y3, y4;   // raw conv outputs
  DevBuf a1, f, a3, eps;   // naive path only: normalised activations
  DevBuf sa, sf;           // Swin variant: convA / convB outputs (256 ch)
  DevBuf bX, bA1, bF, bA3; // fused backward: the four convs' input activations, materialised for the weight gradients
  DevBuf bSa, bSf;         // Swin, split-f16 forward with f16 gradients: the fuse convolutions' raw results as f16 operands
  DevBuf xstash;           // loop backward: the T states entering each step + the running gradient, fp32 NHWC16
  int64_t traj_ticket = 0;         // key.keep plans: ticket of the dd_denoise call whose states x_0 .. x_{T-1} xstash holds (0 = none)
  int64_t traj_weights = -1;       // ... and the parameter generation (dd_handle_s::weights_serial) they were computed with
  bool traj_consumed = false;      // a dd_denoise_backward has read this trajectory: the plan may be dropped when the activation budget is needed
  DevBuf gA, gY;           // backward scratch: gradient w.r.t. a layer's activation / conv output (fp32, up to 256 channels)
  DevBuf dgb;              // backward: per (sample, channel) sums, [B][C][2] (generic kernels) or [B][C][4] (blocked kernels) doubles
  DevBuf bcorr;              // Swin variant, hoisted 5x5 form: this step's border correction [B][swin_ring_size][64] fp32 (swin_bcorr)
  DevBuf ttab, tt_scratch;   // Swin variant, hoisted form: E[t] border tables of the T loop steps [T][SWIN_TT_ROWS][64] (swin_ttab, dd_misc.hip) ...
  int64_t ttab_weights = -1; // ... and the parameter generation they were computed from
  DevBuf ccond;            // Res variant, hoisted condition term: conv3(cond), fp32 in accumulator-fragment order of 8x32 tiles
  DevBuf ccond_raw;        // EK_F16R: the same as the split-f16 layer 8 leaves it (fp32, 8x32 tiles); reformatted into `ccond` (launch_cadd_reformat)
  DevBuf ccond_scale;      // EK_F16R, "f16r_wide": one fp32 scale per accumulator block of the int16 hoisted term (dd_kernels.h)
  DevBuf y3_scale;         // ... and one per pixel of y3 (per-step slots like y3)
  bool wide = false, c1 = false, p4 = false;      // EK_F16R: options "f16r_wide" (y3 / the hoisted term as block-scaled int16, else f16), "f16r_c1" (conv1's
                                                  // weights as an f16 pair, else the plain f16 kernel), "f16r_p4", as this plan was built with them
  DevBuf stats;       // [(T+1)*4][B][STAT_SLOTS][STAT_STRIDE] doubles
  DevBuf c1c2;        // [T][2] fp32
  DevBuf tsteps;      // [T] int64
  std::vector<long long> tsteps_host;   // the same on the host
  size_t stats_bytes = 0;
  hipGraphExec_t exec = nullptr;
  bool capture_failed = false;
  uint64_t last_use = 0;
  ~Plan() { if (exec) (void)hipGraphExecDestroy(exec); }
  int slots = 1;      // per-step copies of y1..y4 / sa / sf (key.keep == 2: T, the backward then recomputes nothing; else 1)
  size_t kept_bytes = 0;   // key.keep == 2: bytes of those per-step slots (what the handle-wide budget "keep_activations_mb" counts)
  void* slot(const DevBuf& b, int step) const { return static_cast<char*>(b.p) + (slots > 1 ? (size_t)step * (b.bytes / slots) : 0); }
  double* stat_ptr(int step, int layer) const {   // layer 0..3
    return stats.as<double>() + ((size_t)(step * 4 + layer) * key.B) * STAT_SLOTS * STAT_STRIDE;
  }
};

// workspace of dd_condition for one pyramid shape / element kind
struct FpnWork {
  int B = 0, ek = -1, pyr = 0, hs[FPN_LEVELS] = {0}, ws[FPN_LEVELS] = {0};
  DevBuf fin[FPN_LEVELS];        // backbone features, activation layout
  DevBuf lat[FPN_LEVELS];        // levels 1..3: x_i = relu(bn(conv(f_i))) [+ top-down term]
  DevBuf up[FPN_LEVELS - 1];     // conv_up[j](x_{j+1}) at 2h x 2w of level j+1
  DevBuf pooled[FPN_LEVELS - 1]; // adaptive_avg_pool2d(up[j]) when 2h_{j+1} x 2w_{j+1} != h_j x w_j
  bool neck = false;             // HAHI neck buffers allocated
  DevBuf nk_cat[FPN_LEVELS];     // channel concatenation the fusion conv reads: [lateral | projection] (level 0: [projection | lateral])
  DevBuf nk_out[FPN_LEVELS];     // neck output = input of the FPN's lateral conv
};


// The HAHI neck's convolutions (reference src/model/necks/hahi.py:60-97; attention off): kernel layer base + 4 * kind + level, base 30 for
// the Swin-L pyramid (192 << level channels), 54 for MPViT-small (128 | 216 | 288 | 288; the kernels carry 216 as 224 channels).
// cout / cin = the reference tensors' sizes; C = real channels of the level, Ck = channels the kernels carry.
struct NeckConv { int layer; std::string name; int cout, cin, ks, level, kind, C, Ck; };
constexpr int NECK_C_MPVIT[4] = {128, 216, 288, 288};
inline int neck_c(int pyr, int level) { return pyr == PYR_MPVIT ? NECK_C_MPVIT[level] : (192 << level); }
inline int neck_ck(int pyr, int level) { return pyr == PYR_MPVIT ? FPN_CIN_MPVIT_PAD[level] : (192 << level); }
inline int neck_base(int pyr) { return pyr == PYR_MPVIT ? 54 : 30; }
inline std::vector<NeckConv> neck_convs(int pyr) {
  std::vector<NeckConv> v;
  const int base = neck_base(pyr);
  for (int i = 0; i < 4; ++i) {
    const int C = neck_c(pyr, i), Ck = neck_ck(pyr, i);
    const std::string si = std::to_string(i), sj = std::to_string(i - 1);
    v.push_back({base + i, "hahineck.lateral_convs." + si, C, C, 1, i, 0, C, Ck});
    v.push_back({base + 4 + i, i == 0 ? std::string("hahineck.conv_proj.0") : "hahineck.trans_proj." + sj, 512, C, 1, i, 1, C, Ck});
    v.push_back({base + 8 + i, i == 0 ? std::string("hahineck.conv_fusion.0") : "hahineck.trans_fusion." + sj, C, C + 512, 3, i, 2, C, Ck});
  }
  return v;
}

}  // namespace ddapi
using namespace ddapi;


struct dd_handle_s {
  static constexpr int MAX_LANES = 4;     // concurrent sub-batches of one dd_denoise / dd_denoise_backward call (option "streams")
  int device = 0;
  int variant = DD_VARIANT_RES;
  std::string err;
  std::map<std::string, std::vector<float>> host_w;
  std::map<std::string, std::unique_ptr<DevBuf>> dev_w;   // dd_set_weight_device: fp32 device copies (denoiser group)
  std::set<std::string> dev_newer;                        // names whose device copy is newer than host_w's (or that have no host copy)
  bool committed = false;      // denoiser group (model.*) packed
  bool codec_committed = false; // codec group (depth_transform.*) packed
  ConvLayer L[4];
  ConvLayer LA, LB;            // Swin variant: upsample_fuse.convA / convB (256->256, no norm)
  DevBuf emb;
  DevBuf etab;               // [EMB_ROWS][10][64] per-tap W3 . E[t] (hoisted time-embedding term of conv3)
  // Swin variant, hoisted 5x5 form (SWIN_PRED5_H, dd_kernels.h): pred.0 o convB as one kernel (fp32 OIHW and the packed images of the one-plane
  // kinds), the tap-pair products of the border correction; built by the first hoisted plan that runs after a parameter update
  DevBuf w5_oihw, w5pack[NUM_WIMG], pairp, kside;
  int64_t w5_weights = -1;
  int swin_w5 = 1;           // option "swin_w5": 1 = the 5x5 form, 0 = convB and pred.0 as two kernels (SWIN_PRED_H)
  DevBuf zero_bias;          // 256 zeros
  int hoist_cond = -1;       // Res variant: conv3(cond) once per image (f16 in the bf16 mode) instead of re-adding cond in conv3's prologue every
                             // step.  -1 = automatic: on in the default bf16 mode (EK_BF16M), where it carries precision (the condition
                             // term never passes through bf16 operands: DESIGN.md section 4), and in the f16 mode; saves 256 B / pixel / step; 0 / 1 = forced
  bool bf16_pure = false;    // option "bf16_storage": DD_PREC_BF16 with all-bf16 tensors and kernels (no f16 anywhere)
  DevBuf codec_buf;          // all folded codec weights in one allocation
  CodecWeights codec{};
  DevBuf codec_tmp;          // scratch for encode/decode intermediates (grown on demand)
  std::vector<float> acp;
  DevBuf d_acp;
  int n_train = 0;
  bool use_graph = true, timing = false, debug_sync = false, layer_timing = false;
  bool graph_off_by_env = false;      // dd_create: the runtime's graph fast path was not switched off in the environment -> eager loops by default
  int graph_fence = 0;                // option "graph_fence" (diagnosis of the training-graph fault, profiles/r06_experiments.md section 5): bit 0 = hipStreamSynchronize in FRONT of every
                                      // hipGraphLaunch, bit 1 = behind it, bit 2 = the graph is launched on the handle's own capture stream between two events instead of on the caller's stream
  bool train_graphs = false;          // option "train_graphs": 1 = the trajectory-keeping forward of a training step replays a hipGraph too (A/B; default eager)
  int check_finite = 0;        // 1 = synchronise after every stage; 2 = count asynchronously, report at the end of the call (no host synchronisation in between)
  DevBuf chk_buf; std::vector<std::string> chk_labels;      // check_finite == 2: one counter per checked stage of the running call   // option "check_finite" (debug): the backward synchronises after every stage and fails with the name of the first tensor holding a NaN / Inf
  int ablate = 0;             // timing experiments only (ConvParams::ablate)
  unsigned long long* prof_buf = nullptr;   // tools/phase_prof.py (-DDD_PHASE_PROF=1 builds): caller-owned device buffer, 8 x u64 per workgroup
  int prof_layer = 0;         // the kernel layer id whose launches write it
  // Training: with option "keep_trajectory" dd_denoise keeps the states entering every step and hands out a ticket (counter
  // "trajectory_ticket"); dd_denoise_backward called after set_option("use_trajectory", ticket) reads them instead of running the forward
  // loop a second time -- if that ticket is still the plan's and the parameters have not changed since, else it regenerates as before.
  bool keep_traj = false;
  int64_t keep_act_mb = 65536;
  int n_streams = 1;          // option "streams": concurrent sub-batches of dd_denoise (1 = off)
  int thin_slots = 512;       // option "thin_slots": workgroups of that kernel (two per CU on the 256 CUs; the tests shrink it to make a workgroup walk several tiles)
  int big_tiles = -1;         // option "big_tiles": hoisted conv3 pair on 16x32 tiles: -1 = when the 8x32 tiles exceed the 512 resident slots, 0 / 1 = forced
  int one_buffer = 1;         // option "one_buffer": the hoisted conv3 on 8x32 tiles in its one-patch-buffer form when the tiles exceed the resident slots (A/B switch)
  int thin_stream = 1;        // option "thin_stream": conv4 as the persistent streaming kernel of dd_thin.hip; 0 = the general kernel (A/B switch)
  int f16r_wide = 1;          // DD_PREC_F16R: y3 and the hoisted conv3(cond) term as block-scaled int16 (0 = as f16, like DD_PREC_F16)
  int f16r_c1 = 1;            // DD_PREC_F16R: conv1's weights as an f16 pair (two MFMAs; 0 = the plain f16 kernel)
  int f16r_p4 = 0;            // DD_PREC_F16R: conv4's operand as an f16 pair as well (two MFMAs per tap)
  bool split_ok = true;       // every forward convolution weight fits the split-f16 images (|w| x 256 inside f16): DD_PREC_F16X3 / DD_PREC_F16R refuse to run otherwise
  DevBuf wmax;                // device route: bits of max |w| over the forward convolution weights (launch_max_abs)
  int resident_slots = 512;   // workgroup slots the chip holds at two per CU (dd_create: 2 x multiProcessorCount): the big-tile rule and thin_slots' default
  hipStream_t lane_stream[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
  std::vector<hipEvent_t> probe_events;
  std::vector<hipStream_t> burnt_streams;      // lane-stream candidates that did not run concurrently with the caller's stream (kept alive: acquire_lane_stream)
  int lane_probe = 1;                          // option "lane_probe": 1 = a lane's stream is probed for concurrency with the caller's when it is created
  int64_t n_lane_probe_retries = 0, lane_overlap_seen = -1;      // counters "lane_probe_retries", "lane_overlap" (-1 = never probed)
  hipEvent_t lane_fork = nullptr, lane_done[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
  int64_t n_lane_calls = 0;
  int active_lanes = 1;       // lanes of the dd_denoise_backward call in progress (the weight-gradient kernels size their slab count by it)
  bool adjoint_tiled = true;  // Swin backward: tiled separable kernel for the adjoint of the condition upsampling (0 = the one-thread-per-piece kernel, A/B check)
  int64_t traj_serial = 0, use_traj = 0, weights_serial = 0, n_traj_reuse = 0;
  int naive_wgrad = 0;        // backward: 1 = weight gradients by the unfused kernel in every mode (A/B check of dd_wgrad.hip)
  int x3_grad_fp32 = 0;       // DD_PREC_F16X3's backward: 0 = f16 gradients behind the split forward (the f16 mode's MFMA kernels), 1 = fp32 gradients (the fp32 mode's kernels)
  std::map<PlanKey, std::unique_ptr<Plan>> plans;
  DevBuf wgrad_ws[MAX_LANES]; // per-slab partial weight gradients of dd_wgrad.hip (one workspace per concurrent lane)
  // parameter gradients (fp32, reference shapes), accumulated like torch .grad in set 0; sets 1.. are the scratch of the concurrent lanes of
  // dd_denoise_backward (added into set 0 and cleared at the join)
  std::map<std::string, std::unique_ptr<DevBuf>> grads[MAX_LANES];
  std::map<std::tuple<int, int, int, int, int>, std::pair<std::shared_ptr<DevBuf>, uint64_t>> cond_bufs;   // (B, h, w, precision, lane) -> buffer, last use
  // condition FPN (Res variant): folded + packed weights, workspace of the last shape, and where its result lives
  bool neck_committed = false;            // hahineck.* folded + packed (DD_VARIANT_SWIN with the Swin-L pyramid only)
  DevBuf neck_w[12][NUM_WIMG], neck_b[12];  // index = kernel layer - 30 (Swin-L pyramid) / - 54 (MPViT-small): 4 * kind + level
  int64_t n_neck_launches = 0;
  bool fpn_committed = false;
  int fpn_pyramid = PYR_DEFAULT;  // PYR_MPVIT once MPViT-sized lateral weights were set (DD_VARIANT_SWIN only)
  DevBuf fpn_lat_w[FPN_LEVELS][NUM_WIMG], fpn_lat_b[FPN_LEVELS];        // (index = image slot: fp32, bf16, f16, WIMG_SPLIT)
  DevBuf fpn_up_w[FPN_LEVELS - 1][NUM_WIMG], fpn_up_b[FPN_LEVELS - 1];
  std::unique_ptr<FpnWork> fpn_work;
  std::shared_ptr<DevBuf> fpn_out;       // Swin: FPN result at the pyramid's finest size (activation layout), upsampled per dd_denoise
  bool fpn_split_ok = false, neck_split_ok = false;   // the folded FPN / neck weights fit the split-f16 images (else those modes run the pyramid on the fp32-operand kernels)
  int cond_direct = 1;                   // option "cond_direct": 0 = the refined f16 mode converts an explicit condition tensor into the blocked layout first (A/B switch)
  int cond_split = 1;                    // option "cond_split": 0 = the split / refined f16 modes run the once-per-image pyramid on the fp32-operand kernels (round 3's route)
  std::shared_ptr<DevBuf> fpn_cond;      // == the cond buffer dd_condition wrote last (valid until the next dd_condition / explicit cond of that shape)
  int fpn_cond_key[4] = {0, 0, 0, -1};   // B, h, w, precision of fpn_cond
  uint64_t tick = 0;
  Plan* last_once_plan = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool ev_valid = false;
  hipStream_t cap_stream = nullptr;   // capture-only stream (torch's default stream is the NULL stream, which cannot capture)
  int64_t n_graph_launches = 0, n_eager_loops = 0, n_capture_failures = 0;
  static constexpr int N_LAYER_SLOTS = 72;   // kernel layer ids run up to 65 (see dd_igemm2_cfg.h)
  double layer_ms[N_LAYER_SLOTS] = {0};     // index = kernel layer id - 1 (1..4 Res, 5..7 Swin fuse, 8..9 hoisted conv3, 10..18 / 24..26 condition FPN, 20..23 dgrad)
  int64_t layer_cnt[N_LAYER_SLOTS] = {0};
  std::vector<std::tuple<int, hipEvent_t, hipEvent_t>> pending_ev;

  int fail(int code, const std::string& m) { err = m; return code; }
};

#define DD_HIP(expr)                                                                              \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess)                                                                         \
      return h->fail(DD_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));              \
  } while (0)

// ---- functions shared between the units (defined in the unit named) ----
namespace ddapi {
struct WeightSpec { std::string name; int64_t numel; };

inline const char* const kConvNames[4] = {"model.noise_embedding.0", "model.noise_embedding.3", "model.pred.0", "model.pred.3"};
inline const char* const kGnNames[4] = {"model.noise_embedding.1", "model.noise_embedding.4", "model.pred.1", "model.pred.4"};
constexpr int kCins[4] = {LATENT_C, HID_C, COND_C, HID_C}, kCouts[4] = {HID_C, COND_C, HID_C, LATENT_C};

// dd_api_weights.cpp
int weight_group(const std::string& name);
std::vector<WeightSpec> required_weights(int variant, int pyr = PYR_DEFAULT);
bool pack_conv_weights(const float* w_oihw, const PackGeom& g, int ek, bool swizzle, std::vector<uint8_t>& out);
int upload(dd_handle_t h, DevBuf& dst, const void* src, size_t bytes, hipStream_t s);
int pull_device_weights_to_host(dd_handle_t h, hipStream_t s);
int ensure_bytes(dd_handle_t h, DevBuf& dst, size_t bytes);
int pack_conv_layer_device(dd_handle_t h, ConvLayer& L, const float* w, int fwd_layer, int dgrad_layer, bool with_naive, hipStream_t s);
int commit_model_from_device(dd_handle_t h, hipStream_t s);
// dd_api_plans.cpp
int check_common(dd_handle_t h, int B, int lh, int lw, int ch, int cw, bool need_schedule);
int lane_count(dd_handle_t h, int B, int precision);
int check_split(dd_handle_t h, int precision, const char* who);
int acquire_lane_stream(dd_handle_t h, int lane, hipStream_t caller);      // dd_api.cpp: creates h->lane_stream[lane] (probed for concurrency with `caller`)
int get_cond_buf(dd_handle_t h, int B, int lh, int lw, int precision, std::shared_ptr<DevBuf>* out, int lane = 0);
int want_hoist(dd_handle_t h, int precision, int T = 1, int keep = 0);
bool keep2_fits(dd_handle_t h, size_t need);
bool plan_big_tiles(dd_handle_t h, const PlanKey& key);
int get_plan(dd_handle_t h, const PlanKey& key, Plan** out);
int enqueue_fused_step(dd_handle_t h, Plan* pl, int step, const float* x_in, float* x_out, bool apply_update,
                       const long long* tvec, int t_base, int t_bstride, hipStream_t s);
int enqueue_naive_eps(dd_handle_t h, Plan* pl, int step, const float* x_in, const long long* tvec, int t_base,
                      int t_bstride, hipStream_t s);
int enqueue_cond_conv(dd_handle_t h, Plan* pl, hipStream_t s, const float* nchw = nullptr);
int ensure_swin_w5(dd_handle_t h, hipStream_t s);
int enqueue_swin_hoist(dd_handle_t h, Plan* pl, hipStream_t s);
int stage_condition(dd_handle_t h, Plan* pl, const float* cond, int B, int lat_h, int lat_w, int cond_h, int cond_w,
                    int precision, hipStream_t s, int img0 = 0, int whole_B = 0);
int enqueue_loop_body(dd_handle_t h, Plan* pl, hipStream_t s);
void drain_layer_events(dd_handle_t h);
// dd_api_train.cpp
float* grad_buf(dd_handle_t h, const std::string& name, size_t numel, hipStream_t s, hipError_t* err, int lane = 0);
int ensure_bwd_buffers(dd_handle_t h, Plan* pl);
int bwd_core(dd_handle_t h, Plan* pl, const float* x_nhwc, const long long* tv, int t_base, int t_bstride, float* grad_cond,
             int accumulate_cond, hipStream_t s, const Plan* kept = nullptr, int kstep = 0, int lane = 0);
int check_bwd(dd_handle_t h, int precision, const char* who);
int denoise_backward_lane(dd_handle_t h, const float* x_T, const float* cond, const float* grad_x0, float* grad_xT, float* grad_cond,
                          int B, int lat_h, int lat_w, int cond_h, int cond_w, int T, int precision, hipStream_t s, int lane, int img0,
                          int whole_B, int64_t ticket, bool* reused, int S);
}  // namespace ddapi
