// dd_api.cpp -- C ABI (include/ddepth.h) over the HIP kernels: handle, parameter packing, per-shape
// plans (scratch + schedule tables + captured hipGraph of the T-step loop) and launch sequencing.
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

namespace {

thread_local std::string g_create_error;

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

}  // namespace

// The HAHI neck's convolutions (reference src/model/necks/hahi.py:60-97; attention off): kernel layer base + 4 * kind + level, base 30 for
// the Swin-L pyramid (192 << level channels), 54 for MPViT-small (128 | 216 | 288 | 288; the kernels carry 216 as 224 channels).
// cout / cin = the reference tensors' sizes; C = real channels of the level, Ck = channels the kernels carry.
struct NeckConv { int layer; std::string name; int cout, cin, ks, level, kind, C, Ck; };
static const int NECK_C_MPVIT[4] = {128, 216, 288, 288};
inline int neck_c(int pyr, int level) { return pyr == PYR_MPVIT ? NECK_C_MPVIT[level] : (192 << level); }
inline int neck_ck(int pyr, int level) { return pyr == PYR_MPVIT ? FPN_CIN_MPVIT_PAD[level] : (192 << level); }
inline int neck_base(int pyr) { return pyr == PYR_MPVIT ? 54 : 30; }
static std::vector<NeckConv> neck_convs(int pyr) {
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
  int thin_xcd = 1;           // option "thin_xcd": that kernel's workgroup -> tile map keeps each XCD on one contiguous block of tiles (halo rows out of its L2); 0 = interleaved (A/B switch)
  int thin_stream = 1;        // option "thin_stream": conv4 as the persistent streaming kernel of dd_thin.hip; 0 = the general kernel (A/B switch)
  int f16r_wide = 1;          // DD_PREC_F16R: y3 and the hoisted conv3(cond) term as block-scaled int16 (0 = as f16, like DD_PREC_F16)
  int f16r_c1 = 1;            // DD_PREC_F16R: conv1's weights as an f16 pair (two MFMAs; 0 = the plain f16 kernel)
  int f16r_p4 = 0;            // DD_PREC_F16R: conv4's operand as an f16 pair as well (two MFMAs per tap)
  bool split_ok = true;       // every forward convolution weight fits the split-f16 images (|w| x 256 inside f16): DD_PREC_F16X3 / DD_PREC_F16R refuse to run otherwise
  DevBuf wmax;                // device route: bits of max |w| over the forward convolution weights (launch_max_abs)
  int resident_slots = 512;   // workgroup slots the chip holds at two per CU (dd_create: 2 x multiProcessorCount): the big-tile rule and thin_slots' default
  hipStream_t lane_stream[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t lane_fork = nullptr, lane_done[MAX_LANES] = {nullptr, nullptr, nullptr, nullptr};
  int64_t n_lane_calls = 0;
  int active_lanes = 1;       // lanes of the dd_denoise_backward call in progress (the weight-gradient kernels size their slab count by it)
  bool adjoint_tiled = true;  // Swin backward: tiled separable kernel for the adjoint of the condition upsampling (0 = the one-thread-per-piece kernel, A/B check)
  int64_t traj_serial = 0, use_traj = 0, weights_serial = 0, n_traj_reuse = 0;
  int naive_wgrad = 0;        // backward: 1 = weight gradients by the unfused kernel in every mode (A/B check of dd_wgrad.hip)
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

namespace {

struct WeightSpec { std::string name; int64_t numel; };

const char* const kConvNames[4] = {"model.noise_embedding.0", "model.noise_embedding.3", "model.pred.0", "model.pred.3"};
const char* const kGnNames[4] = {"model.noise_embedding.1", "model.noise_embedding.4", "model.pred.1", "model.pred.4"};
constexpr int kCins[4] = {LATENT_C, HID_C, COND_C, HID_C}, kCouts[4] = {HID_C, COND_C, HID_C, LATENT_C};

// 0 = denoiser (model.*), 1 = latent codec (depth_transform.*), 2 = condition FPN (conv_lateral.* / conv_up.*),
// 3 = HAHI neck in front of the FPN (hahineck.*; Swin-L pyramid only)
int weight_group(const std::string& name) {
  if (name.compare(0, 6, "model.") == 0) return 0;
  if (name.compare(0, 16, "depth_transform.") == 0) return 1;
  if (name.compare(0, 9, "hahineck.") == 0) return 3;
  return 2;
}

std::vector<WeightSpec> required_weights(int variant, int pyr = PYR_DEFAULT) {
  std::vector<WeightSpec> v = {
      {"model.noise_embedding.0.weight", 64 * 16 * 9}, {"model.noise_embedding.0.bias", 64},
      {"model.noise_embedding.1.weight", 64}, {"model.noise_embedding.1.bias", 64},
      {"model.noise_embedding.3.weight", 256 * 64 * 9}, {"model.noise_embedding.3.bias", 256},
      {"model.noise_embedding.4.weight", 256}, {"model.noise_embedding.4.bias", 256},
      {"model.time_embedding.weight", (int64_t)EMB_ROWS * COND_C},
      {"model.pred.0.weight", 64 * 256 * 9}, {"model.pred.0.bias", 64},
      {"model.pred.1.weight", 64}, {"model.pred.1.bias", 64},
      {"model.pred.3.weight", 16 * 64 * 9}, {"model.pred.3.bias", 16},
      {"model.pred.4.weight", 16}, {"model.pred.4.bias", 16},
      {"depth_transform.conv_transform.0.0.weight", 16 * 9},
      {"depth_transform.conv_transform.0.1.weight", 16}, {"depth_transform.conv_transform.0.1.bias", 16},
      {"depth_transform.conv_transform.0.1.running_mean", 16}, {"depth_transform.conv_transform.0.1.running_var", 16},
      {"depth_transform.conv_transform.1.0.weight", 16 * 16 * 9},
      {"depth_transform.conv_transform.1.1.weight", 16}, {"depth_transform.conv_transform.1.1.bias", 16},
      {"depth_transform.conv_transform.1.1.running_mean", 16}, {"depth_transform.conv_transform.1.1.running_var", 16},
      {"depth_transform.conv_inv_transform.0.weight", 16 * 16 * 16}, {"depth_transform.conv_inv_transform.0.bias", 16},
      {"depth_transform.conv_inv_transform.1.weight", 16}, {"depth_transform.conv_inv_transform.1.bias", 16},
      {"depth_transform.conv_inv_transform.1.running_mean", 16}, {"depth_transform.conv_inv_transform.1.running_var", 16},
      {"depth_transform.conv_inv_transform.3.0.weight", 16 * 9}, {"depth_transform.conv_inv_transform.3.0.bias", 1},
  };
  if (variant == DD_VARIANT_SWIN) {
    v.push_back({"model.upsample_fuse.convA.conv.weight", 256 * 256 * 9});
    v.push_back({"model.upsample_fuse.convA.conv.bias", 256});
    v.push_back({"model.upsample_fuse.convB.conv.weight", 256 * 256 * 9});
    v.push_back({"model.upsample_fuse.convB.conv.bias", 256});
  }
  {
    // condition aggregation FPN of the Res / Swin heads (reference ...res.py:56-84): conv_lateral[i] = Conv3x3(bias=False)+BN+ReLU,
    // conv_up[j] = ConvTranspose2d(k2,s2,bias=False)+BN+ReLU
    const char* bn[4] = {"weight", "bias", "running_mean", "running_var"};
    for (int i = 0; i < FPN_LEVELS; ++i) {
      const std::string pre = "conv_lateral." + std::to_string(i);
      v.push_back({pre + ".0.weight", (int64_t)COND_C * fpn_cin(variant, pyr)[i] * 9});
      for (const char* b : bn) v.push_back({pre + ".1." + b, COND_C});
    }
    for (int j = 0; j < FPN_LEVELS - 1; ++j) {
      const std::string pre = "conv_up." + std::to_string(j);
      v.push_back({pre + ".0.weight", (int64_t)COND_C * COND_C * 4});
      for (const char* b : bn) v.push_back({pre + ".1." + b, COND_C});
    }
  }
  if (variant == DD_VARIANT_SWIN) {
    // HAHI neck (optional 4th group): ConvModule = bias-free conv + BatchNorm + ReLU
    const char* bn[4] = {"weight", "bias", "running_mean", "running_var"};
    for (const NeckConv& c : neck_convs(pyr)) {
      v.push_back({c.name + ".conv.weight", (int64_t)c.cout * c.cin * c.ks * c.ks});
      for (const char* b : bn) v.push_back({c.name + ".bn." + b, c.cout});
    }
  }
  return v;
}

// Packed layout consumed by conv_igemm_kernel:
//   [n_tile][cin_chunk][tap_group][tap_in_group][n (NT)][k (CK)]  of  W[cout][cin][dy][dx]   (zero beyond COUT; ks x ks taps)
// returns false when a weight does not fit the split-f16 image (|w| x SPLIT_WSCALE beyond f16): dd_commit_weights records it in
// dd_handle_s::split_ok and the split modes (DD_PREC_F16X3 / DD_PREC_F16R) refuse to run on such parameters -- the other precisions are unaffected
bool pack_conv_weights(const float* w_oihw, const PackGeom& g, int ek, bool swizzle, std::vector<uint8_t>& out) {
  bool fits = true;
  const int ks = g.ks, n_tiles = g.cout_pad / g.nt, n_chunks = g.cin / g.ck, n_tg = ks * ks / g.tg;
  const int planes = g.planes > 1 ? 2 : 1;       // split f16 (EK_F16S): every stage block is [hi plane | lo plane] of f16 elements
  const size_t n_el = (size_t)n_tiles * n_chunks * n_tg * g.tg * g.nt * g.ck * planes;
  const size_t esz = planes == 2 ? 2 : ek_size(ek);
  const size_t plane_el = (size_t)g.tg * g.nt * g.ck;
  out.assign(n_el * esz, 0);
  for (int nt = 0; nt < n_tiles; ++nt)
    for (int ch = 0; ch < n_chunks; ++ch)
      for (int tg = 0; tg < n_tg; ++tg)
        for (int t = 0; t < g.tg; ++t) {
          const int tap = tg * g.tg + t, dy = tap / ks, dx = tap % ks;
          const int rowb = g.ck * (int)esz, ppp = rowb / 16, rpb = 256 / rowb, epp = 16 / (int)esz;
          const size_t blk0 = (((size_t)(nt * n_chunks + ch) * n_tg + tg) * g.tg + 0) * (size_t)g.nt * g.ck * planes;
          for (int n = 0; n < g.nt; ++n)
            for (int k = 0; k < g.ck; ++k) {
              const int co = nt * g.nt + n, ci = ch * g.ck + k;
              // element index inside the packed image; v2 XORs the 16-B piece index with the row swizzle
              const int row = t * g.nt + n;
              const int piece = k / epp, within = k % epp;
              const int piece_sw = swizzle ? (piece ^ ((row / rpb) & (ppp - 1))) : piece;
              const size_t idx = blk0 + (size_t)row * g.ck + (size_t)piece_sw * epp + within;
              const float v = (co < g.cout) ? w_oihw[(((size_t)co * g.cin + ci) * ks + dy) * ks + dx] : 0.f;
              if (g.stack && co >= g.cout && co < 2 * g.cout) {
                // stacked image (conv4, EK_F16R): cout row cout + c = the lo half of row c times STACK_LSCALE (pack_weights_kernel's arithmetic)
                const float wv = w_oihw[(((size_t)(co - g.cout) * g.cin + ci) * ks + dy) * ks + dx];
                if (!(std::fabs(wv) < 60000.f)) fits = false;
                const uint16_t u = host_f32_to_f16((wv - (float)(_Float16)wv) * STACK_LSCALE);
                std::memcpy(&out[idx * 2], &u, 2);
                continue;
              }
              if (planes == 2) {
                // hi = f16(w * SPLIT_WSCALE), lo = f16(w * SPLIT_WSCALE - hi): the same arithmetic as pack_weights_kernel (dd_misc.hip)
                const float vs = v * SPLIT_WSCALE;
                if (!(std::fabs(vs) < 60000.f)) fits = false;       // |w| >= 234: beyond f16 after scaling (reported by the caller)
                const _Float16 hi = (_Float16)vs;
                const _Float16 lo = (_Float16)(vs - (float)hi);
                std::memcpy(&out[idx * 2], &hi, 2);
                std::memcpy(&out[(idx + plane_el) * 2], &lo, 2);
              } else if (ek == EK_F32) std::memcpy(&out[idx * 4], &v, 4);
              else {
                const uint16_t u = (ek == EK_BF16) ? host_f32_to_bf16(v) : host_f32_to_f16(v);
                std::memcpy(&out[idx * 2], &u, 2);
              }
            }
        }
  return fits;
}

int upload(dd_handle_t h, DevBuf& dst, const void* src, size_t bytes, hipStream_t s) {
  if (dst.bytes < bytes || !dst.p) DD_HIP(dst.alloc(bytes));
  DD_HIP(hipMemcpyAsync(dst.p, src, bytes, hipMemcpyHostToDevice, s));
  return DD_OK;
}

// need_schedule: the T-step loop (and its backward) reads the DDIM tables; ONE epsilon-network evaluation does not
// (reference ...res.py:324-344 has no scheduler dependency), so dd_denoise_once / _backward run on a handle without a schedule.
int check_common(dd_handle_t h, int B, int lh, int lw, int ch, int cw, bool need_schedule) {
  if (!h) return DD_ERR_INVALID_ARG;
  if (!h->committed) return h->fail(DD_ERR_STATE, "model.* weights not committed (call dd_set_weight for every key, then dd_commit_weights)");
  if (need_schedule && h->n_train <= 0) return h->fail(DD_ERR_STATE, "schedule not set (dd_set_schedule)");
  if (B <= 0 || lh <= 0 || lw <= 0) return h->fail(DD_ERR_INVALID_ARG, "B, lat_h, lat_w must be positive");
  if (h->variant == DD_VARIANT_RES && (ch != lh || cw != lw))
    return h->fail(DD_ERR_INVALID_ARG, "DD_VARIANT_RES needs cond_h,cond_w == lat_h,lat_w (reference ...res.py:340 adds them elementwise)");
  if (h->variant == DD_VARIANT_SWIN && (ch <= 0 || cw <= 0)) return h->fail(DD_ERR_INVALID_ARG, "cond_h, cond_w must be positive");
  if ((long long)B * lh * lw * COND_C >= (1LL << 31) * 4) return h->fail(DD_ERR_INVALID_ARG, "tensor too large");
  return DD_OK;
}

// Concurrent lanes of ONE dd_denoise / dd_denoise_backward call (option "streams"): the same rule for the forward and the backward, so that a
// backward always looks for the kept trajectory under the keys the forward stored it (one stream for the unfused path, the per-launch timing
// mode, debug synchronisation and the phase profiler: their per-call state -- pending_ev, prof_buf -- is not per lane).
int lane_count(dd_handle_t h, int B, int precision) {
  int S = h->n_streams;
  if (S > B) S = B;
  if (S > dd_handle_s::MAX_LANES) S = dd_handle_s::MAX_LANES;
  if (precision == DD_PREC_NAIVE_FP32 || h->layer_timing || h->debug_sync || h->prof_buf) S = 1;
  return S < 1 ? 1 : S;
}

// The split modes' preconditions: parameters that fit the split-f16 images (dd_commit_weights records it, for both routes); DD_PREC_F16R is built
// for the Res denoiser
int check_split(dd_handle_t h, int precision, const char* who) {
  if (precision != DD_PREC_F16X3 && precision != DD_PREC_F16R) return DD_OK;
  if (!h->split_ok)
    return h->fail(DD_ERR_UNSUPPORTED, std::string(who) + ": a convolution weight of magnitude >= 234 does not fit the split-f16 images (weights are scaled by 256 "
                                       "into f16): DD_PREC_F16X3 / DD_PREC_F16R cannot run on these parameters; the other precisions can");
  return DD_OK;
}

// The condition map at latent size in the activation layout of `precision`: one buffer per (B, h, w, precision), shared by
// all plans of that shape (graphs bake its address) and written in place by dd_condition.
int get_cond_buf(dd_handle_t h, int B, int lh, int lw, int precision, std::shared_ptr<DevBuf>* out, int lane = 0) {
  const auto key = std::make_tuple(B, lh, lw, precision, lane);
  auto it = h->cond_bufs.find(key);
  if (it == h->cond_bufs.end()) {
    while (h->cond_bufs.size() >= 12) {           // plans keep their buffer alive through the shared_ptr
      auto victim = h->cond_bufs.begin();
      for (auto j = h->cond_bufs.begin(); j != h->cond_bufs.end(); ++j)
        if (j->second.second < victim->second.second) victim = j;
      if (h->fpn_cond == victim->second.first) { h->fpn_cond.reset(); h->fpn_cond_key[3] = -1; }
      h->cond_bufs.erase(victim);
    }
    auto buf = std::make_shared<DevBuf>();
    const size_t es = precision == DD_PREC_NAIVE_FP32 ? 4 : ek_size(cond_kind(ek_of_precision(precision, h->bf16_pure)));
    DD_HIP(buf->alloc((size_t)B * lh * lw * COND_C * es));
    it = h->cond_bufs.emplace(key, std::make_pair(buf, (uint64_t)0)).first;
  }
  it->second.second = ++h->tick;
  *out = it->second.first;
  return DD_OK;
}

// conv3's condition term out of the loop?  (Res variant, fused modes; option "hoist_cond": -1 = in the 16-bit modes whose tensors are
// stored in f16 -- the default bf16 mode and the f16 mode: faster AND closer to the fp32 path, the condition term reaching the accumulators
// in fp32 (f16 mode at KITTI size: 533 vs 502 maps/s, depth RMSE 1.54e-4 vs 1.74e-4); the fp32 parity mode keeps the reference's order of sums)
// Swin variant (SWIN_CONVA_H / SWIN_PRED_H, dd_kernels.h): the whole step-invariant part of pred.0(convB(convA(.))) -- condition map and
// time embedding through three convolutions -- in the plans of the loop that keep nothing for a backward (T > 0, keep == 0): the weight
// gradients of convB / pred.0 need the un-split activations, so training plans and the single-call plans (per-sample timesteps) run the
// reference's order.  -1 = in the 2-byte modes and the split-f16 mode; 1 also in the fp32 mode.
int want_hoist(dd_handle_t h, int precision, int T = 1, int keep = 0) {
  if (precision == DD_PREC_NAIVE_FP32) return 0;
  const int ek = ek_of_precision(precision, h->bf16_pure);
  if (h->variant == DD_VARIANT_SWIN) {
    if (T <= 0 || keep != 0 || h->hoist_cond == 0 || ((ek == EK_F16S || ek == EK_F16R) && !h->swin_w5)) return 0;      // (split / refined f16: the 5x5 form only)
    return (h->hoist_cond == 1 || ek != EK_F32) ? 1 : 0;
  }
  if (h->variant != DD_VARIANT_RES) return 0;
  if (ek == EK_F16R) return 1;           // the mode IS the hoisted form: its condition map exists as fp32 for the once-per-image split conv3 only
  if (h->hoist_cond >= 0) return h->hoist_cond;
  return (ek == EK_BF16M || ek == EK_F16 || ek == EK_F16S) ? 1 : 0;
}

// May a NEW plan keep `need` bytes of per-step activations (PlanKey::keep == 2)?  The budget (option "keep_activations_mb", default 64 GiB)
// is ONE figure for the handle -- all lanes, all shapes -- and is also held against the HBM that is actually free.  Plans whose kept
// trajectory nobody can ask for any more (ticket consumed by its backward, or invalidated by a parameter update / a newer forward) are
// dropped first, least recently used first: a training loop that alternates shapes (train / validation crops) neither accumulates one
// activation set per shape nor falls back to recompute while stale sets sit in HBM.
bool keep2_fits(dd_handle_t h, size_t need) {
  const size_t cap = (size_t)h->keep_act_mb << 20;
  if (need > cap) return false;
  for (;;) {
    size_t held = 0;
    for (auto& kv : h->plans) held += kv.second->kept_bytes;
    size_t free_b = 0, total_b = 0;
    const bool mem_ok = hipMemGetInfo(&free_b, &total_b) != hipSuccess || need + (need >> 3) <= free_b;    // 12 % headroom for the rest of the step
    if (held + need <= cap && mem_ok) return true;
    auto victim = h->plans.end();
    for (auto j = h->plans.begin(); j != h->plans.end(); ++j) {
      const Plan& q = *j->second;
      if (q.kept_bytes == 0 || (q.traj_ticket != 0 && q.traj_weights == h->weights_serial && !q.traj_consumed)) continue;      // a live trajectory: its backward is still to come
      if (victim == h->plans.end() || q.last_use < victim->second->last_use) victim = j;
    }
    if (victim == h->plans.end()) return false;
    if (h->last_once_plan == victim->second.get()) h->last_once_plan = nullptr;
    (void)hipDeviceSynchronize();
    h->plans.erase(victim);
  }
}

// The hoisted conv3 pair of a plan runs on 16x32-pixel tiles (kernel ids BIG_CONV3C / BIG_CONV3H, dd_kernels.h) when its 8x32 tiles would not
// fit the chip's resident workgroup slots at once; a function of the plan key only, so that the once-per-image kernel, the loop kernel, the
// buffer of the hoisted term and every backward recompute agree.  Option "big_tiles": -1 = this rule, 0 / 1 = forced (A/B, tests).
bool plan_big_tiles(dd_handle_t h, const PlanKey& key) {
  if (!key.hoist || key.prec == DD_PREC_NAIVE_FP32) return false;
  if (h->variant != DD_VARIANT_RES && !(h->variant == DD_VARIANT_SWIN && h->swin_w5)) return false;      // Swin: the 5x5 form and its once-per-image layer 8
  const int ek = ek_of_precision(key.prec, h->bf16_pure);
  if (ek == EK_F32 || ek == EK_F16S) return false;
  if (h->big_tiles >= 0) return h->big_tiles != 0;
  // more 8x32 tiles than resident slots: 16x32 tiles under concurrent lanes (half the weight stream and 0.75 LDS reads per MFMA: what counts when
  // the other lane keeps the chip full anyway); a call that runs as ONE lane keeps the 8x32 tiles in their one-patch-buffer form (three workgroups
  // per CU: conv3 140 -> 132 us at KITTI B=4, profiles/r04_call3_*) -- the Res denoiser's conv3; the Swin 5x5 form has no such kernel
  const bool many = (long long)key.B * ((key.h + 7) / 8) * ((key.w + 31) / 32) > h->resident_slots;
  return many && (key.lanes > 1 || h->variant == DD_VARIANT_SWIN);
}
inline int conv3c_kid(dd_handle_t h, const PlanKey& key) { return plan_big_tiles(h, key) ? (int)BIG_CONV3C : 8; }
// the loop's hoisted conv3: 16x32 tiles, or 8x32 tiles -- with ONE patch buffer (kernel id ONE_CONV3H: same tiles, same fragment order, 52 KB
// of LDS = three workgroups per CU) when there are more tiles than the chip holds at two per CU
inline int conv3h_kid(dd_handle_t h, const PlanKey& key) {
  if (plan_big_tiles(h, key)) return (int)BIG_CONV3H;
  const bool many = (long long)key.B * ((key.h + 7) / 8) * ((key.w + 31) / 32) > h->resident_slots;
  return (h->one_buffer == 2 || (many && h->one_buffer)) ? (int)ONE_CONV3H : 9;      // (2 = always: tests)
}

int get_plan(dd_handle_t h, const PlanKey& key, Plan** out) {
  auto it = h->plans.find(key);
  if (it != h->plans.end()) { it->second->last_use = ++h->tick; *out = it->second.get(); return DD_OK; }
  // keep at most 10 plans alive: evict the least recently used
  // (a plan whose kept trajectory is still owed a backward -- ticket unconsumed, parameters unchanged -- is not a victim, as in keep2_fits:
  // evicting it would silently turn that backward into a recompute; with lanes a training loop holds keep + backward plans per lane)
  const size_t cap = 10 + 2 * (size_t)(h->n_streams > 1 ? (h->n_streams < dd_handle_s::MAX_LANES ? h->n_streams : dd_handle_s::MAX_LANES) : 0);
  while (h->plans.size() >= cap) {
    auto victim = h->plans.end();
    for (auto j = h->plans.begin(); j != h->plans.end(); ++j) {
      const Plan& q = *j->second;
      if (q.key.keep && q.traj_ticket != 0 && q.traj_weights == h->weights_serial && !q.traj_consumed) continue;
      if (victim == h->plans.end() || q.last_use < victim->second->last_use) victim = j;
    }
    if (victim == h->plans.end()) break;           // every plan holds a live trajectory: grow rather than lose one
    if (h->last_once_plan == victim->second.get()) h->last_once_plan = nullptr;
    DD_HIP(hipDeviceSynchronize());
    h->plans.erase(victim);
  }
  std::unique_ptr<Plan> pl(new Plan());
  pl->key = key;
  const bool naive = key.prec == DD_PREC_NAIVE_FP32;
  pl->ek = naive ? EK_F32 : ek_of_precision(key.prec, h->bf16_pure);
  if (pl->ek == EK_F16R && !key.hoist)
    return h->fail(DD_ERR_UNSUPPORTED, "DD_PREC_F16R runs the hoisted forward-only plans: for the Swin / MPViT denoiser that is the T-step loop (dd_denoise / "
                                       "dd_denoise_trace) with option swin_w5 = 1; single calls (dd_denoise_once) and training plans: DD_PREC_F16");
  const size_t px = (size_t)key.B * key.h * key.w;
  const size_t es = ek_size(pl->ek);
  DD_HIP(pl->x[0].alloc(px * LATENT_C * 4));
  DD_HIP(pl->x[1].alloc(px * LATENT_C * 4));
  if (key.keep) DD_HIP(pl->xstash.alloc((size_t)(key.T > 0 ? key.T : 1) * px * LATENT_C * 4));
  const bool swin = h->variant == DD_VARIANT_SWIN;
  // Swin: the condition map is bilinearly upsampled to the latent size once per call and kept at that size
  (void)swin;
  // (refined f16, Res denoiser: hoisted forward-only plans whose loop never reads the condition map, and whose once-per-image conv3(cond) reads an
  // explicit `cond` tensor in place (option "cond_direct"): the blocked fp32 buffer -- 219 MB at KITTI B = 4 -- is only allocated when a call needs
  // it: a non-direct stage_condition, or dd_condition's resident map)
  if (!(pl->ek == EK_F16R && !swin && key.hoist)) { int rc = get_cond_buf(h, key.B, key.h, key.w, key.prec, &pl->cond, key.lane); if (rc) return rc; }
  pl->slots = (key.keep == 2 && key.T > 0) ? key.T : 1;
  const size_t ns = (size_t)pl->slots;
  if (swin) {      // (refined f16: the once-per-image chain runs on split operands through fp32 tensors in these two buffers)
    const size_t es_s = pl->ek == EK_F16R ? 4 : es;
    DD_HIP(pl->sa.alloc(ns * px * COND_C * es_s)); DD_HIP(pl->sf.alloc(ns * px * COND_C * es_s));
  }
  if (key.hoist)   // conv3(cond) in accumulator-fragment order: whole (th x 32)-pixel tiles
  {
    const int th = conv_pack_geom2(conv3h_kid(h, key), pl->ek).th;
    DD_HIP(pl->ccond.alloc((size_t)key.B * ((key.h + th - 1) / th) * ((key.w + 31) / 32) * th * 32 * HID_C * 4));
    if (pl->ek == EK_F16R) {
      pl->wide = h->f16r_wide != 0; pl->c1 = h->f16r_c1 != 0; pl->p4 = h->f16r_p4 != 0;
      DD_HIP(pl->ccond_raw.alloc((size_t)key.B * ((key.h + 7) / 8) * ((key.w + 31) / 32) * 8 * 32 * HID_C * 4));      // what the split layer 8 writes
      if (pl->wide) DD_HIP(pl->ccond_scale.alloc((size_t)key.B * ((key.h + th - 1) / th) * ((key.w + 31) / 32) * 4 * 2 * (th / 4) * 4));
    }
  }
  if (key.hoist && swin) {
    const int T1 = key.T > 0 ? key.T : 1;
    const int RH = key.h < SWIN_TT_AX ? key.h : SWIN_TT_AX, RW = key.w < SWIN_TT_AX ? key.w : SWIN_TT_AX;
    DD_HIP(pl->ttab.alloc((size_t)T1 * SWIN_TT_ROWS * HID_C * 4));
    DD_HIP(pl->tt_scratch.alloc((size_t)T1 * RH * RW * (2 * COND_C + HID_C) * 4));
    if (h->swin_w5) {
      DD_HIP(pl->bcorr.alloc((size_t)key.B * swin_ring_stride(key.h, key.w) * HID_C * 4));
      DD_HIP(hipMemsetAsync(pl->bcorr.p, 0, pl->bcorr.bytes, nullptr));
      DD_HIP(hipStreamSynchronize(nullptr));
    }
  }
  DD_HIP(pl->y1.alloc(ns * px * HID_C * es));
  DD_HIP(pl->y2.alloc(ns * px * COND_C * es));
  DD_HIP(pl->y3.alloc(ns * px * HID_C * es));
  if (pl->ek == EK_F16R && h->f16r_wide) DD_HIP(pl->y3_scale.alloc(ns * px * 4));
  DD_HIP(pl->y4.alloc(ns * px * LATENT_C * 4));
  if (key.keep == 2) pl->kept_bytes = pl->y1.bytes + pl->y2.bytes + pl->y3.bytes + pl->y3_scale.bytes + pl->y4.bytes + pl->sa.bytes + pl->sf.bytes;
  if (naive) {
    DD_HIP(pl->a1.alloc(px * HID_C * 4));
    DD_HIP(pl->f.alloc(px * COND_C * 4));
    DD_HIP(pl->a3.alloc(px * HID_C * 4));
    DD_HIP(pl->eps.alloc(px * LATENT_C * 4));
  }
  const int T = key.T > 0 ? key.T : 1;
  pl->stats_bytes = (size_t)(T + 1) * 4 * key.B * STAT_SLOTS * STAT_STRIDE * sizeof(double);
  DD_HIP(pl->stats.alloc(pl->stats_bytes));
  // schedule tables (reference scheduling_ddim.py:215-229 timesteps, :285-326 closed form of step())
  std::vector<float> c1c2((size_t)T * 2, 0.f);
  std::vector<long long> ts((size_t)T, 0);
  if (key.T > 0) {
    const int ratio = h->n_train / key.T;
    for (int k = 0; k < key.T; ++k) {
      const int t = (key.T - 1 - k) * ratio;
      const int prev = t - ratio;
      const double a_t = (double)h->acp[t];
      const double a_prev = prev >= 0 ? (double)h->acp[prev] : 1.0;      // final_alpha_cumprod (set_alpha_to_one)
      c1c2[2 * k] = (float)std::sqrt(a_prev / a_t);
      c1c2[2 * k + 1] = (float)(std::sqrt(1.0 - a_prev) - std::sqrt(a_prev * (1.0 - a_t) / a_t));
      ts[k] = t;
    }
  }
  DD_HIP(pl->c1c2.alloc(c1c2.size() * 4));
  DD_HIP(pl->tsteps.alloc(ts.size() * 8));
  pl->tsteps_host = ts;
  DD_HIP(hipMemcpy(pl->c1c2.p, c1c2.data(), c1c2.size() * 4, hipMemcpyHostToDevice));
  DD_HIP(hipMemcpy(pl->tsteps.p, ts.data(), ts.size() * 8, hipMemcpyHostToDevice));
  pl->last_use = ++h->tick;
  *out = pl.get();
  h->plans[key] = std::move(pl);
  return DD_OK;
}

// One epsilon-network evaluation of the fused path: conv1..conv4 at loop step `step`
// (x_in -> [update] -> conv1 ... conv4 -> y4 + GN4 statistics in stat slot `step`).
int enqueue_fused_step(dd_handle_t h, Plan* pl, int step, const float* x_in, float* x_out, bool apply_update,
                       const long long* tvec, int t_base, int t_bstride, hipStream_t s) {
  const PlanKey& k = pl->key;
  ConvParams p{};
  p.B = k.B; p.h = k.h; p.w = k.w;
  p.tiles_x = (k.w + 31) / 32;
  p.tiles_y = (k.h + 7) / 8;
  p.ablate = h->ablate;
  // the loop's timesteps are the plan's own schedule: pass the value, not the address (clamped as clamp_t does on the device)
  if (tvec == pl->tsteps.as<long long>() && t_bstride == 0 && t_base >= 0 && t_base < (int)pl->tsteps_host.size())
    p.t_known = clamp_t(pl->tsteps_host[t_base]);
  const int ek = pl->ek, tk = thin_kind(ek);    // mode; kind of conv1 / conv4
  const bool rf = ek == EK_F16R;                // refined f16 (dd_kernels.h): split conv1, f16 conv2 / conv3 (fp32 hand-over when pl->wide), stacked conv4
  const int wk = ek == EK_F16S ? WIMG_SPLIT : opnd_kind(ek);    // weight image of the large convolutions (their operand kind; the split image in the split mode)
  // conv4 runs as the persistent streaming kernel of dd_thin.hip in the 2-byte modes (option "thin_stream", default on; the phase profiler
  // instruments the general kernel)
  const bool stream4 = h->thin_stream && (tk == EK_F16 || tk == EK_BF16) && !h->prof_buf && k.B <= h->thin_slots;
  auto timed_launch = [&](int layer, const ConvParams& cp, int kid_as = -1) -> hipError_t {
    const int kid = kid_as >= 0 ? kid_as : layer == 9 ? conv3h_kid(h, k) : layer;            // kernel id; times are booked under `layer`
    auto launch = [&](ConvParams q) {
      q.prof = (h->prof_buf && layer == h->prof_layer) ? h->prof_buf : nullptr;
      q.tiles_y = (k.h + conv_pack_geom2(kid, ek).th - 1) / conv_pack_geom2(kid, ek).th;
      if (layer == 4 && rf) { q.persist_slots = h->thin_slots; q.xcd_map = h->thin_xcd; q.cadd_scale = static_cast<const float*>(pl->slot(pl->y3_scale, step)); return launch_conv4_stream(EK_F16, q, s, true, pl->wide, pl->p4); }
      if (layer == 4 && stream4) { q.persist_slots = h->thin_slots; q.xcd_map = h->thin_xcd; return launch_conv4_stream(tk, q, s); }
      int lek = ek;
      if (rf && (layer == 9 || layer == 7) && !pl->wide) lek = EK_F16;      // hand-over of y3 / the hoisted term as f16: the f16 mode's conv3 / 5x5 form
      if (rf && layer == 1 && !pl->c1) lek = EK_F16;        // conv1 without the weight pair: the f16 mode's conv1
      return launch_conv_igemm2(kid, lek, q, s);
    };
    if (!h->layer_timing) return launch(cp);
    hipEvent_t a, b;
    hipError_t e = hipEventCreate(&a); if (e != hipSuccess) return e;
    e = hipEventCreate(&b); if (e != hipSuccess) return e;
    (void)hipEventRecord(a, s);
    e = launch(cp);
    (void)hipEventRecord(b, s);
    h->pending_ev.emplace_back(layer - 1, a, b);
    return e;
  };
  // this step's activation buffers (per-step slots in the plans that keep them for the backward)
  void *y1_ = pl->slot(pl->y1, step), *y2_ = pl->slot(pl->y2, step), *y3_ = pl->slot(pl->y3, step), *y4_ = pl->slot(pl->y4, step);
  void *sa_ = pl->slot(pl->sa, step), *sf_ = pl->slot(pl->sf, step);
  const float* y4_prev = static_cast<const float*>(pl->slot(pl->y4, step > 0 ? step - 1 : 0));     // read by the fused update of step - 1
  // conv1: state (+ fused DDIM update of the previous step) -> y1
  p.in = x_in; p.wpack = h->L[0].wpack2[rf ? (pl->c1 ? WIMG_SPLIT : (int)EK_F16) : wimg_slot(tk)].p; p.bias = h->L[0].bias.as<float>(); p.out = y1_;
  p.stats_out = pl->stat_ptr(step, 0);
  p.stats_in = apply_update ? pl->stat_ptr(step - 1, 3) : nullptr;
  p.gn_gamma = h->L[3].gamma.as<float>(); p.gn_beta = h->L[3].beta.as<float>();
  p.y4 = y4_prev; p.xout = x_out; p.c1c2 = pl->c1c2.as<float>(); p.step = apply_update ? step : 0;
  DD_HIP(timed_launch(1, p));
  // conv2: relu(gn1(y1)) -> y2
  p.in = y1_; p.wpack = h->L[1].wpack2[wk].p; p.bias = h->L[1].bias.as<float>(); p.out = y2_;
  p.stats_out = pl->stat_ptr(step, 1); p.stats_in = pl->stat_ptr(step, 0);
  p.gn_gamma = h->L[0].gamma.as<float>(); p.gn_beta = h->L[0].beta.as<float>();
  DD_HIP(timed_launch(2, p));
  if (h->variant == DD_VARIANT_SWIN && k.hoist) {
    // hoisted form: pred.0(convB(convA(relu(gn2(y2))))) without the fuse convs' biases; the accumulators of pred.0 start at the per-image
    // term of enqueue_swin_hoist and its epilogue adds this step's E[t] rows
    p.in = y2_; p.wpack = h->LA.wpack2[wk].p; p.bias = h->zero_bias.as<float>(); p.out = sa_;
    p.stats_out = nullptr; p.stats_in = pl->stat_ptr(step, 1);
    p.gn_gamma = h->L[1].gamma.as<float>(); p.gn_beta = h->L[1].beta.as<float>();
    DD_HIP(timed_launch(5, p, SWIN_CONVA_H));
    p.stats_in = nullptr;
    p.cadd = pl->ccond.as<float>(); p.ttab = pl->ttab.as<float>() + (size_t)step * SWIN_TT_ROWS * HID_C;
    if (pl->bcorr.p) {
      // pred.0 o convB as one 5x5 convolution on convA's result; the border ring's correction first
      DD_HIP(launch_swin_bcorr(sa_, ek == EK_F16S ? (int)EK_F32 : opnd_kind(ek), h->pairp.as<float>(), h->kside.p, pl->bcorr.as<float>(), k.B, k.h, k.w, s));     // (kind convA' stored its result in)
      p.in = sa_; p.wpack = h->w5pack[wk].p; p.bias = h->L[2].bias.as<float>(); p.out = y3_;
      p.stats_out = pl->stat_ptr(step, 2); p.bcorr = pl->bcorr.as<float>();
      p.cadd_scale = pl->ccond_scale.as<float>(); p.out_scale = static_cast<float*>(pl->slot(pl->y3_scale, step));      // (EK_F16R form only)
      DD_HIP(timed_launch(7, p, plan_big_tiles(h, k) ? (int)SWIN_PRED5B_H : (int)SWIN_PRED5_H));
    } else {
      p.in = sa_; p.wpack = h->LB.wpack2[wk].p; p.out = sf_;
      DD_HIP(timed_launch(6, p));
      p.in = sf_; p.wpack = h->L[2].wpack2[wk].p; p.bias = h->L[2].bias.as<float>(); p.out = y3_;
      p.stats_out = pl->stat_ptr(step, 2);
      DD_HIP(timed_launch(7, p, SWIN_PRED_H));
    }
  } else if (h->variant == DD_VARIANT_SWIN) {
    // upsample_fuse: convB(convA(relu(gn2(y2)) + up(cond) + E[t]))  then pred.0 on the raw result
    p.in = y2_; p.wpack = h->LA.wpack2[wk].p; p.bias = h->LA.bias.as<float>(); p.out = sa_;
    p.stats_out = nullptr; p.stats_in = pl->stat_ptr(step, 1);
    p.gn_gamma = h->L[1].gamma.as<float>(); p.gn_beta = h->L[1].beta.as<float>();
    p.cond = pl->cond_ptr(); p.emb = h->emb.as<float>(); p.tvec = tvec; p.t_base = t_base; p.t_bstride = t_bstride;
    DD_HIP(timed_launch(5, p));
    p.in = sa_; p.wpack = h->LB.wpack2[wk].p; p.bias = h->LB.bias.as<float>(); p.out = sf_;
    p.stats_in = nullptr;
    DD_HIP(timed_launch(6, p));
    p.in = sf_; p.wpack = h->L[2].wpack2[wk].p; p.bias = h->L[2].bias.as<float>(); p.out = y3_;
    p.stats_out = pl->stat_ptr(step, 2);
    DD_HIP(timed_launch(7, p));
  } else {
  // conv3: relu(gn2(y2)) + cond + E[t] -> y3   (hoisted form: conv3(relu(gn2(y2))) + [conv3(cond) + conv3(E[t])])
  p.in = y2_; p.wpack = h->L[2].wpack2[wk].p; p.bias = h->L[2].bias.as<float>(); p.out = y3_;
  p.stats_out = pl->stat_ptr(step, 2); p.stats_in = pl->stat_ptr(step, 1);
  p.gn_gamma = h->L[1].gamma.as<float>(); p.gn_beta = h->L[1].beta.as<float>();
  p.cond = pl->cond_ptr(); p.emb = h->emb.as<float>(); p.tvec = tvec; p.t_base = t_base; p.t_bstride = t_bstride;
  p.cadd = pl->ccond.as<float>(); p.etab = h->etab.as<float>();
  p.cadd_scale = pl->ccond_scale.as<float>(); p.out_scale = static_cast<float*>(pl->slot(pl->y3_scale, step));      // (EK_F16R forms only)
  DD_HIP(timed_launch(k.hoist ? 9 : 3, p));
  p.cadd_scale = nullptr; p.out_scale = nullptr;
  }
  // conv4: relu(gn3(y3)) -> y4 (fp32)
  p.in = y3_; p.wpack = h->L[3].wpack2[rf ? WIMG_STACK : wimg_slot(tk)].p; p.bias = h->L[3].bias.as<float>(); p.out = y4_;
  p.stats_out = pl->stat_ptr(step, 3); p.stats_in = pl->stat_ptr(step, 2);
  p.gn_gamma = h->L[2].gamma.as<float>(); p.gn_beta = h->L[2].beta.as<float>();
  DD_HIP(timed_launch(4, p));
  if (h->debug_sync) DD_HIP(hipStreamSynchronize(s));
  return DD_OK;
}

// The naive (unfused fp32) epsilon network; leaves eps in pl->eps and raw outputs in y1..y4.
int enqueue_naive_eps(dd_handle_t h, Plan* pl, int step, const float* x_in, const long long* tvec, int t_base,
                      int t_bstride, hipStream_t s) {
  const PlanKey& k = pl->key;
  const int B = k.B, hh = k.h, ww = k.w;
  auto L = h->L;
  DD_HIP(launch_naive_conv3x3(x_in, L[0].w_oihw.as<float>(), L[0].bias.as<float>(), pl->y1.as<float>(), B, hh, ww, LATENT_C, HID_C, s));
  DD_HIP(launch_naive_gn_stats(pl->y1.as<float>(), pl->stat_ptr(step, 0), B, hh, ww, HID_C, s));
  DD_HIP(launch_naive_gn_apply(pl->y1.as<float>(), pl->stat_ptr(step, 0), L[0].gamma.as<float>(), L[0].beta.as<float>(),
                               nullptr, nullptr, nullptr, 0, 0, pl->a1.as<float>(), B, hh, ww, HID_C, s));
  DD_HIP(launch_naive_conv3x3(pl->a1.as<float>(), L[1].w_oihw.as<float>(), L[1].bias.as<float>(), pl->y2.as<float>(), B, hh, ww, HID_C, COND_C, s));
  DD_HIP(launch_naive_gn_stats(pl->y2.as<float>(), pl->stat_ptr(step, 1), B, hh, ww, COND_C, s));
  DD_HIP(launch_naive_gn_apply(pl->y2.as<float>(), pl->stat_ptr(step, 1), L[1].gamma.as<float>(), L[1].beta.as<float>(),
                               static_cast<const float*>(pl->cond_ptr()), h->emb.as<float>(), tvec, t_base, t_bstride, pl->f.as<float>(), B, hh, ww, COND_C, s));
  DD_HIP(launch_naive_conv3x3(pl->f.as<float>(), L[2].w_oihw.as<float>(), L[2].bias.as<float>(), pl->y3.as<float>(), B, hh, ww, COND_C, HID_C, s));
  DD_HIP(launch_naive_gn_stats(pl->y3.as<float>(), pl->stat_ptr(step, 2), B, hh, ww, HID_C, s));
  DD_HIP(launch_naive_gn_apply(pl->y3.as<float>(), pl->stat_ptr(step, 2), L[2].gamma.as<float>(), L[2].beta.as<float>(),
                               nullptr, nullptr, nullptr, 0, 0, pl->a3.as<float>(), B, hh, ww, HID_C, s));
  DD_HIP(launch_naive_conv3x3(pl->a3.as<float>(), L[3].w_oihw.as<float>(), L[3].bias.as<float>(), pl->y4.as<float>(), B, hh, ww, HID_C, LATENT_C, s));
  DD_HIP(launch_naive_gn_stats(pl->y4.as<float>(), pl->stat_ptr(step, 3), B, hh, ww, LATENT_C, s));
  DD_HIP(launch_naive_gn_apply(pl->y4.as<float>(), pl->stat_ptr(step, 3), L[3].gamma.as<float>(), L[3].beta.as<float>(),
                               nullptr, nullptr, nullptr, 0, 0, pl->eps.as<float>(), B, hh, ww, LATENT_C, s));
  if (h->debug_sync) DD_HIP(hipStreamSynchronize(s));
  return DD_OK;
}

// conv3 applied once to the (already converted) condition map: the per-image part of the hoisted conv3 (layer 8)
// nchw != nullptr (refined f16 only): the caller's NCHW fp32 tensor read in place by the kernel id CONV3C_NCHW -- no channel-blocked copy exists
int enqueue_cond_conv(dd_handle_t h, Plan* pl, hipStream_t s, const float* nchw = nullptr) {
  const PlanKey& k = pl->key;
  ConvParams p{};
  p.B = k.B; p.h = k.h; p.w = k.w;
  p.tiles_x = (k.w + 31) / 32;
  p.ablate = 0;
  if (pl->ek == EK_F16R) {
    // refined f16: the split-f16 layer 8 on the fp32 condition map (the term is exact to ~22 bits), fp32 in the order of 8x32 tiles; then into
    // the order / element type the loop's conv3 reads (block-scaled int16 or f16 quads, 8x32 or 16x32 tiles)
    p.tiles_y = (k.h + 7) / 8;
    p.in = nchw ? static_cast<const void*>(nchw) : pl->cond_ptr(); p.wpack = h->L[2].wpack2[WIMG_SPLIT].p; p.bias = h->zero_bias.as<float>();
    p.out = pl->ccond_raw.p;
    DD_HIP(launch_conv_igemm2(nchw ? CONV3C_NCHW : 8, EK_F16S, p, s));
    DD_HIP(launch_cadd_reformat(pl->ccond_raw.as<float>(), pl->ccond.p, pl->ccond_scale.as<float>(), k.B, k.h, k.w, plan_big_tiles(h, k) ? 1 : 0, pl->wide ? 2 : 1, s));
    return DD_OK;
  }
  const int kid = conv3c_kid(h, k);
  p.tiles_y = (k.h + conv_pack_geom2(kid, pl->ek).th - 1) / conv_pack_geom2(kid, pl->ek).th;
  p.in = pl->cond_ptr(); p.wpack = h->L[2].wpack2[wimg_slot(thin_kind(pl->ek))].p; p.bias = h->zero_bias.as<float>(); p.out = pl->ccond.p;
  DD_HIP(launch_conv_igemm2(kid, pl->ek, p, s));
  return DD_OK;
}

int ensure_bytes(dd_handle_t h, DevBuf& dst, size_t bytes);

// Swin variant, hoisted 5x5 form: pred.0 o convB of the current parameter generation -- fp32 composition, tap-pair products and line kernels of the
// border correction, packed images of the one-plane kinds -- on stream s.  Handle-wide data: dd_denoise calls this on the caller's stream BEFORE it
// forks its lanes (a lane that found it stale would rebuild it on its own stream under the other lanes' feet).
int ensure_swin_w5(dd_handle_t h, hipStream_t s) {
  if (h->w5_weights == h->weights_serial) return DD_OK;
  if (!h->LB.w_oihw.p || !h->L[2].w_oihw.p) return h->fail(DD_ERR_STATE, "hoisted Swin form: the fp32 weights of convB / pred.0 are not on the device");
  if (!h->w5_oihw.p) {
    DD_HIP(h->w5_oihw.alloc((size_t)HID_C * COND_C * 25 * 4)); DD_HIP(h->pairp.alloc((size_t)81 * COND_C * HID_C * 4)); DD_HIP(h->kside.alloc(SWIN_KSIDE_BYTES));
  }
  DD_HIP(launch_swin_compose(h->LB.w_oihw.as<float>(), h->L[2].w_oihw.as<float>(), h->w5_oihw.as<float>(), h->pairp.as<float>(), h->kside.p, s));
  for (int wi = 0; wi < NUM_WIMG; ++wi) {
    if (wi == WIMG_STACK) continue;
    const int ekk = wimg_kind(wi);
    const PackGeom g5 = conv_pack_geom2(SWIN_PRED5_H, ekk);
    int rc = ensure_bytes(h, h->w5pack[wi], pack_weights_bytes(g5, ekk)); if (rc) return rc;
    DD_HIP(launch_pack_weights(h->w5_oihw.as<float>(), h->w5pack[wi].p, g5, ekk, true, false, s));
  }
  h->w5_weights = h->weights_serial;
  return DD_OK;
}

// Swin variant, hoisted form: the per-image term pred.0(convB(convA(up(feat)) + a) + b) without pred.0's bias, left in the accumulator-fragment
// order of pred.0's tiles (layer 8), and -- once per plan and parameter generation -- the E[t] tables of the loop's steps.  The upsampled
// condition map is in the plan's buffer; Plan::sa / sf are free until the loop starts.
int enqueue_swin_hoist(dd_handle_t h, Plan* pl, hipStream_t s) {
  const PlanKey& k = pl->key;
  if (pl->ttab_weights != h->weights_serial) {
    if (!h->LA.w_oihw.p || !h->LB.w_oihw.p || !h->L[2].w_oihw.p) return h->fail(DD_ERR_STATE, "hoisted Swin form: the fp32 weights of the fuse convolutions are not on the device");
    DD_HIP(launch_swin_ttab(h->LA.w_oihw.as<float>(), h->LB.w_oihw.as<float>(), h->L[2].w_oihw.as<float>(), h->emb.as<float>(),
                            pl->tsteps.as<long long>(), k.T, k.h, k.w, pl->tt_scratch.as<float>(), pl->ttab.as<float>(), s));
    pl->ttab_weights = h->weights_serial;
  }
  if (pl->bcorr.p) { int rc = ensure_swin_w5(h, s); if (rc) return rc; }
  ConvParams p{};
  p.B = k.B; p.h = k.h; p.w = k.w;
  p.tiles_x = (k.w + 31) / 32;
  if (pl->ek == EK_F16R) {
    // refined f16: the whole once-per-image chain on split operands (EK_F16S kernels, fp32 tensors: the upsampled condition map is fp32 in this
    // mode), its result in the order of 8x32 tiles, then reformatted into what the loop's 5x5 kernel reads (as the Res variant's conv3(cond))
    p.tiles_y = (k.h + 7) / 8;
    p.in = pl->cond_ptr(); p.wpack = h->LA.wpack2[WIMG_SPLIT].p; p.bias = h->LA.bias.as<float>(); p.out = pl->sa.p;
    DD_HIP(launch_conv_igemm2(6, EK_F16S, p, s));
    p.in = pl->sa.p; p.wpack = h->LB.wpack2[WIMG_SPLIT].p; p.bias = h->LB.bias.as<float>(); p.out = pl->sf.p;
    DD_HIP(launch_conv_igemm2(6, EK_F16S, p, s));
    p.in = pl->sf.p; p.wpack = h->L[2].wpack2[WIMG_SPLIT].p; p.bias = h->zero_bias.as<float>(); p.out = pl->ccond_raw.p;
    DD_HIP(launch_conv_igemm2(8, EK_F16S, p, s));
    DD_HIP(launch_cadd_reformat(pl->ccond_raw.as<float>(), pl->ccond.p, pl->ccond_scale.as<float>(), k.B, k.h, k.w, plan_big_tiles(h, k) ? 1 : 0, pl->wide ? 2 : 1, s));
    return DD_OK;
  }
  const int tk = thin_kind(pl->ek);           // once per image: f16 kernels in the bf16 mode, as the Res variant's conv3(cond)
  p.tiles_y = (k.h + conv_pack_geom2(6, tk).th - 1) / conv_pack_geom2(6, tk).th;
  p.in = pl->cond_ptr(); p.wpack = h->LA.wpack2[wimg_slot(tk)].p; p.bias = h->LA.bias.as<float>(); p.out = pl->sa.p;
  DD_HIP(launch_conv_igemm2(6, tk, p, s));
  p.in = pl->sa.p; p.wpack = h->LB.wpack2[wimg_slot(tk)].p; p.bias = h->LB.bias.as<float>(); p.out = pl->sf.p;
  DD_HIP(launch_conv_igemm2(6, tk, p, s));
  const int kid8 = conv3c_kid(h, k);          // (16x32 tiles when the loop's pred.0 kernel runs on them: the two agree on the fragment order)
  p.tiles_y = (k.h + conv_pack_geom2(kid8, pl->ek).th - 1) / conv_pack_geom2(kid8, pl->ek).th;
  p.in = pl->sf.p; p.wpack = h->L[2].wpack2[wimg_slot(tk)].p; p.bias = h->zero_bias.as<float>(); p.out = pl->ccond.p;
  DD_HIP(launch_conv_igemm2(kid8, pl->ek, p, s));
  return DD_OK;
}

// Bring the condition map into the plan's (shared) buffer: convert the caller's NCHW fp32 tensor, or -- cond == NULL --
// check that dd_condition left its result there.
// A lane (img0 > 0 or B < whole_B) is handed its images of the caller's tensor by the caller; of dd_condition's result it takes its slice.
int stage_condition(dd_handle_t h, Plan* pl, const float* cond, int B, int lat_h, int lat_w, int cond_h, int cond_w,
                    int precision, hipStream_t s, int img0 = 0, int whole_B = 0) {
  if (whole_B <= 0) whole_B = B;
  pl->cond_alias = nullptr;
  // Refined f16, Res denoiser: its plans are hoisted forward-only ones -- the loop never reads the condition map, only the once-per-image
  // conv3(cond) does, and that kernel can read the caller's NCHW tensor in place (option "cond_direct", default on): no blocked copy is made.
  const bool direct = cond && h->cond_direct && h->variant != DD_VARIANT_SWIN && pl->ek == EK_F16R && pl->key.hoist;
  if (direct) return enqueue_cond_conv(h, pl, s, cond);     // (the plan's blocked buffer -- possibly dd_condition's resident map -- is left alone)
  if (!pl->cond) { int rc = get_cond_buf(h, pl->key.B, pl->key.h, pl->key.w, pl->key.prec, &pl->cond, pl->key.lane); if (rc) return rc; }      // allocated on first need (get_plan)
  if (cond) {
    if (h->variant == DD_VARIANT_SWIN) DD_HIP(launch_upsample_to_blocked(cond, pl->cond->p, cond_kind(pl->ek), B, COND_C, cond_h, cond_w, lat_h, lat_w, s));
    else DD_HIP(launch_nchw_to_nhwc(cond, pl->cond->p, cond_kind(pl->ek), B, COND_C, cond_h, cond_w, precision != DD_PREC_NAIVE_FP32, s));
    if (h->fpn_cond == pl->cond) { h->fpn_cond.reset(); h->fpn_cond_key[3] = -1; }    // overwritten
  } else if (h->variant == DD_VARIANT_SWIN) {
    const int* k = h->fpn_cond_key;
    if (!h->fpn_cond || h->fpn_cond != h->fpn_out || k[0] != whole_B || k[1] != cond_h || k[2] != cond_w || k[3] != precision)
      return h->fail(DD_ERR_STATE, "cond == NULL needs a preceding dd_condition with the same batch, condition size and precision");
    const char* src = static_cast<const char*>(h->fpn_out->p) + (size_t)img0 * COND_C * cond_h * cond_w * ek_size(cond_kind(pl->ek));
    DD_HIP(launch_upsample_blocked(src, pl->cond->p, cond_kind(pl->ek), B, COND_C, cond_h, cond_w, lat_h, lat_w, s));
  } else {
    const int* k = h->fpn_cond_key;
    const bool whole = img0 == 0 && B == whole_B;
    if (!h->fpn_cond || (whole && h->fpn_cond != pl->cond) || k[0] != whole_B || k[1] != lat_h || k[2] != lat_w || k[3] != precision)
      return h->fail(DD_ERR_STATE, "cond == NULL needs a preceding dd_condition with the same batch, latent size and precision");
    if (!whole)     // this lane's images inside the whole batch's condition map (per-image contiguous in the activation layout)
      pl->cond_alias = static_cast<const char*>(h->fpn_cond->p) + (size_t)img0 * lat_h * lat_w * COND_C * ek_size(cond_kind(pl->ek));
  }
  if (pl->key.hoist) { int rc = h->variant == DD_VARIANT_SWIN ? enqueue_swin_hoist(h, pl, s) : enqueue_cond_conv(h, pl, s); if (rc) return rc; }
  return DD_OK;
}

int enqueue_loop_body(dd_handle_t h, Plan* pl, hipStream_t s) {
  const int T = pl->key.T;
  DD_HIP(hipMemsetAsync(pl->stats.p, 0, pl->stats_bytes, s));
  for (int k = 0; k < T; ++k) {
    // conv1 of step k applies the update of step k-1: reads x[(k-1)&1] (k>0) / x[0] (k=0), writes x[k&1]
    const float* xin = (k == 0) ? pl->x[0].as<float>() : pl->x[(k - 1) & 1].as<float>();
    float* xout = pl->x[k & 1].as<float>();
    if (pl->key.keep) {      // X[k] = state entering step k, all T of them kept (X[0] is the input: step 0 writes nothing)
      const size_t n16 = (size_t)pl->key.B * pl->key.h * pl->key.w * LATENT_C;
      float* X = pl->xstash.as<float>();
      xin = (k == 0) ? X : X + (size_t)(k - 1) * n16;
      xout = X + (size_t)k * n16;
    }
    int rc = enqueue_fused_step(h, pl, k, xin, xout, k > 0, pl->tsteps.as<long long>(), k, 0, s);
    if (rc != DD_OK) return rc;
  }
  return DD_OK;
}

void drain_layer_events(dd_handle_t h) {
  for (auto& t : h->pending_ev) {
    float ms = 0.f;
    if (hipEventSynchronize(std::get<2>(t)) == hipSuccess && hipEventElapsedTime(&ms, std::get<1>(t), std::get<2>(t)) == hipSuccess) {
      const int slot = std::get<0>(t);
      if (slot >= 0 && slot < dd_handle_s::N_LAYER_SLOTS) {
        h->layer_ms[slot] += ms;
        h->layer_cnt[slot] += 1;
      }
    }
    (void)hipEventDestroy(std::get<1>(t));
    (void)hipEventDestroy(std::get<2>(t));
  }
  h->pending_ev.clear();
}

}  // namespace

// =================================================================================================
extern "C" {

const char* dd_version(void) { return "ddepth 0.1 gfx950 (hip, mfma bf16/f16/f32 implicit-GEMM conv3x3+GN+ReLU, hipGraph DDIM loop)"; }

const char* dd_last_error(dd_handle_t h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int dd_create(dd_handle_t* out, int device, int variant) {
  if (!out) { g_create_error = "dd_create: out is NULL"; return DD_ERR_INVALID_ARG; }
  *out = nullptr;
  if (variant != DD_VARIANT_RES && variant != DD_VARIANT_SWIN) { g_create_error = "dd_create: unknown variant"; return DD_ERR_INVALID_ARG; }
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) {
    g_create_error = std::string("dd_create: no HIP device available (") + (e != hipSuccess ? hipGetErrorString(e) : "count = 0") +
                     "); this library has no CPU fallback";
    return DD_ERR_HIP;
  }
  if (device < 0 || device >= n) { g_create_error = "dd_create: device index out of range"; return DD_ERR_INVALID_ARG; }
  e = hipSetDevice(device);
  if (e != hipSuccess) { g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e); return DD_ERR_HIP; }
  dd_handle_t h = new dd_handle_s();
  h->device = device;
  h->variant = variant;
  {
    // two workgroups per CU is what the persistent conv4 fills and what the big-tile rule compares tile counts with: from the device, not
    // from "256 CUs" (partitioned modes -- CPX / NPS -- and other parts expose other counts); the options override
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) {
      h->resident_slots = 2 * prop.multiProcessorCount;
      h->thin_slots = h->resident_slots > 4096 ? 4096 : h->resident_slots;
    }
  }
  *out = h;
  return DD_OK;
}

int dd_destroy(dd_handle_t h) {
  if (!h) return DD_OK;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  drain_layer_events(h);
  h->plans.clear();
  if (h->ev0) (void)hipEventDestroy(h->ev0);
  if (h->ev1) (void)hipEventDestroy(h->ev1);
  if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
  for (int l = 0; l < dd_handle_s::MAX_LANES; ++l) {
    if (h->lane_stream[l]) (void)hipStreamDestroy(h->lane_stream[l]);
    if (h->lane_done[l]) (void)hipEventDestroy(h->lane_done[l]);
  }
  if (h->lane_fork) (void)hipEventDestroy(h->lane_fork);
  delete h;
  return DD_OK;
}

int dd_set_weight(dd_handle_t h, const char* name, const float* data, int64_t numel) {
  if (!h) return DD_ERR_INVALID_ARG;
  if (!name || !data || numel <= 0) return h->fail(DD_ERR_INVALID_ARG, "dd_set_weight: null name/data or non-positive numel");
  if (h->variant == DD_VARIANT_SWIN) {
    // the Swin-L and the MPViT-small heads share this variant's denoiser; which pyramid the FPN has shows in the lateral weights
    const std::string nm(name);
    for (int i = 0; i < FPN_LEVELS; ++i)
      if (nm == "conv_lateral." + std::to_string(i) + ".0.weight") {
        const int pyr = numel == (int64_t)COND_C * FPN_CIN_MPVIT[i] * 9 ? PYR_MPVIT : PYR_DEFAULT;
        if (pyr != h->fpn_pyramid) {
          for (int j = 0; j < FPN_LEVELS; ++j) h->host_w.erase("conv_lateral." + std::to_string(j) + ".0.weight");   // other pyramid's
          for (const NeckConv& c : neck_convs(h->fpn_pyramid)) h->host_w.erase(c.name + ".conv.weight");
          h->fpn_pyramid = pyr;
          h->fpn_committed = false;
          h->neck_committed = false;
        }
      }
    // ... or in the neck's first lateral convolution, whichever arrives first (the heads register the neck in front of the FPN)
    if (nm == "hahineck.lateral_convs.0.conv.weight") {
      const int pyr = numel == (int64_t)NECK_C_MPVIT[0] * NECK_C_MPVIT[0] ? PYR_MPVIT : PYR_DEFAULT;
      if (pyr != h->fpn_pyramid) {
        for (int j = 0; j < FPN_LEVELS; ++j) h->host_w.erase("conv_lateral." + std::to_string(j) + ".0.weight");
        for (const NeckConv& c : neck_convs(h->fpn_pyramid)) h->host_w.erase(c.name + ".conv.weight");
        h->fpn_pyramid = pyr;
        h->fpn_committed = false;
        h->neck_committed = false;
      }
    }
  }
  for (const auto& ws : required_weights(h->variant, h->fpn_pyramid)) {
    if (ws.name == name) {
      if (ws.numel != numel)
        return h->fail(DD_ERR_INVALID_ARG, std::string("dd_set_weight: ") + name + " expects " + std::to_string(ws.numel) +
                                               " elements, got " + std::to_string(numel));
      h->host_w[name].assign(data, data + numel);
      h->dev_newer.erase(name);
      h->weights_serial++;
      const int grp = weight_group(name);
      if (grp == 0) h->committed = false; else if (grp == 1) h->codec_committed = false; else if (grp == 3) h->neck_committed = false; else h->fpn_committed = false;
      return DD_OK;
    }
  }
  return h->fail(DD_ERR_INVALID_ARG, std::string("dd_set_weight: unknown parameter name '") + name + "'");
}

int dd_set_weight_device(dd_handle_t h, const char* name, const float* data, int64_t numel, void* stream) {
  if (!h) return DD_ERR_INVALID_ARG;
  if (!name || !data || numel <= 0) return h->fail(DD_ERR_INVALID_ARG, "dd_set_weight_device: null name/data or non-positive numel");
  if (weight_group(name) != 0)
    return h->fail(DD_ERR_UNSUPPORTED, std::string("dd_set_weight_device: '") + name + "' is not a denoiser parameter (model.*): the codec and "
                                       "FPN groups are folded on the host, use dd_set_weight");
  for (const auto& ws : required_weights(h->variant, h->fpn_pyramid)) {
    if (ws.name != name) continue;
    if (ws.numel != numel)
      return h->fail(DD_ERR_INVALID_ARG, std::string("dd_set_weight_device: ") + name + " expects " + std::to_string(ws.numel) +
                                             " elements, got " + std::to_string(numel));
    DD_HIP(hipSetDevice(h->device));
    std::unique_ptr<DevBuf>& b = h->dev_w[name];
    if (!b) b.reset(new DevBuf());
    if (b->bytes != (size_t)numel * 4) DD_HIP(b->alloc((size_t)numel * 4));
    DD_HIP(hipMemcpyAsync(b->p, data, (size_t)numel * 4, hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream)));
    h->dev_newer.insert(name);
    h->committed = false;
    h->weights_serial++;
    return DD_OK;
  }
  return h->fail(DD_ERR_INVALID_ARG, std::string("dd_set_weight_device: unknown parameter name '") + name + "'");
}

namespace {

// host_w <- the device copies that are newer (mixed host / device updates of one group, and the host-only consumers: Winograd images)
int pull_device_weights_to_host(dd_handle_t h, hipStream_t s) {
  if (h->dev_newer.empty()) return DD_OK;
  DD_HIP(hipStreamSynchronize(s));
  for (const std::string& name : h->dev_newer) {
    const DevBuf& b = *h->dev_w[name];
    std::vector<float>& v = h->host_w[name];
    v.resize(b.bytes / 4);
    DD_HIP(hipMemcpy(v.data(), b.p, b.bytes, hipMemcpyDeviceToHost));
  }
  h->dev_newer.clear();
  return DD_OK;
}

int ensure_bytes(dd_handle_t h, DevBuf& dst, size_t bytes) {
  if (dst.bytes < bytes || !dst.p) DD_HIP(dst.alloc(bytes));
  return DD_OK;
}

// One convolution's weights from a device fp32 OIHW tensor into every layout the forward / backward kernels read -- the device twin of the
// host loops in dd_commit_weights (same geometries, same buffers), all on stream `s`.
int pack_conv_layer_device(dd_handle_t h, ConvLayer& L, const float* w, int fwd_layer, int dgrad_layer, bool with_naive, hipStream_t s) {
  for (int wi = 0; wi < NUM_WIMG; ++wi) {
    if (!wimg_has(wi, fwd_layer)) continue;
    const int ek = wimg_kind(wi);
    const PackGeom g2 = conv_pack_geom2(fwd_layer, ek);
    int rc = ensure_bytes(h, L.wpack2[wi], pack_weights_bytes(g2, ek)); if (rc) return rc;
    DD_HIP(launch_pack_weights(w, L.wpack2[wi].p, g2, ek, true, false, s));
    if (wi == WIMG_SPLIT || wi == WIMG_STACK) continue;           // the split / refined f16 modes are forward only
    const PackGeom gt = conv_pack_geom2(dgrad_layer, ek);
    rc = ensure_bytes(h, L.wpackT[ek], pack_weights_bytes(gt, ek)); if (rc) return rc;
    DD_HIP(launch_pack_weights(w, L.wpackT[ek].p, gt, ek, true, true, s));
  }
  if (with_naive) {
    const size_t bytes = (size_t)L.cout * L.cin * 9 * 4;
    int rc = ensure_bytes(h, L.w_oihw, bytes); if (rc) return rc;
    rc = ensure_bytes(h, L.wT_oihw, bytes); if (rc) return rc;
    DD_HIP(hipMemcpyAsync(L.w_oihw.p, w, bytes, hipMemcpyDeviceToDevice, s));
    DD_HIP(launch_transpose_flip(w, L.wT_oihw.as<float>(), L.cout, L.cin, 9, s));
  }
  return DD_OK;
}

// The denoiser group when every one of its parameters came through dd_set_weight_device: no host copy is touched.
int commit_model_from_device(dd_handle_t h, hipStream_t s) {
  auto D = [&](const std::string& n) { return h->dev_w[n]->as<float>(); };
  auto copy_small = [&](DevBuf& dst, const std::string& n, size_t pad_elems) -> int {
    const DevBuf& src = *h->dev_w[n];
    const size_t bytes = std::max(src.bytes, pad_elems * 4);
    int rc = ensure_bytes(h, dst, bytes); if (rc) return rc;
    if (bytes > src.bytes) DD_HIP(hipMemsetAsync(dst.p, 0, bytes, s));
    DD_HIP(hipMemcpyAsync(dst.p, src.p, src.bytes, hipMemcpyDeviceToDevice, s));
    return DD_OK;
  };
  for (int l = 0; l < 4; ++l) {
    ConvLayer& L = h->L[l];
    L.cin = kCins[l]; L.cout = kCouts[l];
    int rc = pack_conv_layer_device(h, L, D(std::string(kConvNames[l]) + ".weight"), l + 1, 23 - l, true, s); if (rc) return rc;
    rc = copy_small(L.bias, std::string(kConvNames[l]) + ".bias", 32); if (rc) return rc;
    rc = copy_small(L.gamma, std::string(kGnNames[l]) + ".weight", 0); if (rc) return rc;
    rc = copy_small(L.beta, std::string(kGnNames[l]) + ".bias", 0); if (rc) return rc;
  }
  if (h->variant == DD_VARIANT_SWIN) {
    const char* names[2] = {"model.upsample_fuse.convA.conv", "model.upsample_fuse.convB.conv"};
    ConvLayer* Ls[2] = {&h->LA, &h->LB};
    for (int i = 0; i < 2; ++i) {
      ConvLayer& L = *Ls[i];
      L.cin = COND_C; L.cout = COND_C;
      int rc = pack_conv_layer_device(h, L, D(std::string(names[i]) + ".weight"), 5 + i, 6, false, s); if (rc) return rc;
      rc = ensure_bytes(h, L.w_oihw, (size_t)COND_C * COND_C * 9 * 4); if (rc) return rc;      // fp32 OIHW: the hoisted form's E[t] tables / 5x5 composition
      DD_HIP(hipMemcpyAsync(L.w_oihw.p, D(std::string(names[i]) + ".weight"), (size_t)COND_C * COND_C * 9 * 4, hipMemcpyDeviceToDevice, s));
      rc = copy_small(L.bias, std::string(names[i]) + ".bias", 0); if (rc) return rc;
    }
  }
  int rc = copy_small(h->emb, "model.time_embedding.weight", 0); if (rc) return rc;
  if (h->etab.bytes == 0) {
    DD_HIP(h->etab.alloc((size_t)EMB_ROWS * 10 * HID_C * 4));
    DD_HIP(h->zero_bias.alloc(COND_C * 4));
    DD_HIP(hipMemsetAsync(h->zero_bias.p, 0, COND_C * 4, s));
  }
  DD_HIP(launch_etab(h->L[2].w_oihw.as<float>(), h->emb.as<float>(), h->etab.as<float>(), s));
  // do the forward weights fit the split-f16 images?  (the host route checks while packing; here: one max-|w| reduction per tensor)
  if (!h->wmax.p) DD_HIP(h->wmax.alloc(sizeof(unsigned)));
  DD_HIP(hipMemsetAsync(h->wmax.p, 0, sizeof(unsigned), s));
  for (int l = 0; l < 4; ++l)
    DD_HIP(launch_max_abs(D(std::string(kConvNames[l]) + ".weight"), (long long)kCouts[l] * kCins[l] * 9, h->wmax.as<unsigned>(), s));
  if (h->variant == DD_VARIANT_SWIN) {
    DD_HIP(launch_max_abs(D("model.upsample_fuse.convA.conv.weight"), (long long)COND_C * COND_C * 9, h->wmax.as<unsigned>(), s));
    DD_HIP(launch_max_abs(D("model.upsample_fuse.convB.conv.weight"), (long long)COND_C * COND_C * 9, h->wmax.as<unsigned>(), s));
  }
  unsigned wbits = 0;
  DD_HIP(hipMemcpyAsync(&wbits, h->wmax.p, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  DD_HIP(hipStreamSynchronize(s));     // as the host route: the new images are in place when the call returns, whatever stream runs next
  float wmax_f;
  std::memcpy(&wmax_f, &wbits, 4);
  h->split_ok = wmax_f * SPLIT_WSCALE < 60000.f;       // (false for NaN)
  return DD_OK;
}

}  // namespace

int dd_commit_weights(dd_handle_t h, void* stream) {
  if (!h) return DD_ERR_INVALID_ARG;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  DD_HIP(hipSetDevice(h->device));
  // independent groups: "model." (denoiser), "depth_transform." (codec) and "conv_lateral." / "conv_up." (condition FPN,
  // Res variant).  A group is packed when all of its keys are present; a partially provided group is an error; at least
  // one must be complete.
  int have[4] = {0, 0, 0, 0}, need[4] = {0, 0, 0, 0}, n_dev = 0, n_host = 0;     // n_dev / n_host: denoiser parameters whose newest value is on the device / host
  std::string first_missing[4];
  for (const auto& ws : required_weights(h->variant, h->fpn_pyramid)) {
    const int grp = weight_group(ws.name);
    need[grp]++;
    if (h->host_w.count(ws.name) || h->dev_newer.count(ws.name)) have[grp]++;
    else if (first_missing[grp].empty()) first_missing[grp] = ws.name;
    if (grp == 0) { if (h->dev_newer.count(ws.name)) n_dev++; else n_host++; }
  }
  for (int grp = 0; grp < 4; ++grp)
    if (have[grp] != 0 && have[grp] != need[grp])
      return h->fail(DD_ERR_STATE, "dd_commit_weights: missing parameter '" + first_missing[grp] + "'");
  if (have[0] == 0 && have[1] == 0 && have[2] == 0 && have[3] == 0) return h->fail(DD_ERR_STATE, "dd_commit_weights: no parameters were set");
  // only the groups that changed since their last commit (dd_set_weight / dd_set_weight_device clear the group's flag)
  bool do_model = have[0] == need[0] && !h->committed;
  const bool do_codec = have[1] == need[1] && !h->codec_committed, do_fpn = need[2] > 0 && have[2] == need[2] && !h->fpn_committed;
  const bool do_neck = need[3] > 0 && have[3] == need[3] && !h->neck_committed;
  if (!do_model && !do_codec && !do_fpn && !do_neck) return DD_OK;
  // graphs bake weight pointers; buffers are reused when sizes match, so existing graphs stay valid,
  // but make sure nothing is in flight while we overwrite them.
  DD_HIP(hipDeviceSynchronize());
  if (do_model && n_dev > 0) {
    if (n_host == 0) {
      // every denoiser parameter came through dd_set_weight_device: fp32 -> kernel layouts by the pack kernels, all on `s`
      int rc = commit_model_from_device(h, s); if (rc) return rc;
      h->committed = true;
      do_model = false;
    } else {
      int rc = pull_device_weights_to_host(h, s); if (rc) return rc;            // mixed update: newest values to the host, host route
    }
  }
  bool split_fits = true;       // every forward weight fits the split-f16 image (checked while packing, reported once the group is through)
  const char* conv_names[4] = {"model.noise_embedding.0", "model.noise_embedding.3", "model.pred.0", "model.pred.3"};
  const char* gn_names[4] = {"model.noise_embedding.1", "model.noise_embedding.4", "model.pred.1", "model.pred.4"};
  const int cins[4] = {LATENT_C, HID_C, COND_C, HID_C}, couts[4] = {HID_C, COND_C, HID_C, LATENT_C};
  for (int l = 0; do_model && l < 4; ++l) {
    ConvLayer& L = h->L[l];
    L.cin = cins[l]; L.cout = couts[l];
    const std::vector<float>& w = h->host_w[std::string(conv_names[l]) + ".weight"];
    const std::vector<float>& b = h->host_w[std::string(conv_names[l]) + ".bias"];
    for (int wi = 0; wi < NUM_WIMG; ++wi) {
      if (!wimg_has(wi, l + 1)) continue;
      std::vector<uint8_t> packed;
      if (!pack_conv_weights(w.data(), conv_pack_geom2(l + 1, wimg_kind(wi)), wimg_kind(wi), true, packed)) split_fits = false;
      int rc = upload(h, L.wpack2[wi], packed.data(), packed.size(), s);
      if (rc) return rc;
      DD_HIP(hipStreamSynchronize(s));     // `packed` is a temporary
    }
    std::vector<float> bpad(std::max(32, L.cout), 0.f);
    std::copy(b.begin(), b.end(), bpad.begin());
    int rc = upload(h, L.bias, bpad.data(), bpad.size() * 4, s); if (rc) return rc;
    rc = upload(h, L.w_oihw, w.data(), w.size() * 4, s); if (rc) return rc;

    {
      std::vector<float> wt(w.size());
      for (int co = 0; co < L.cout; ++co)
        for (int ci = 0; ci < L.cin; ++ci)
          for (int k = 0; k < 9; ++k) wt[((size_t)ci * L.cout + co) * 9 + (8 - k)] = w[((size_t)co * L.cin + ci) * 9 + k];
      rc = upload(h, L.wT_oihw, wt.data(), wt.size() * 4, s); if (rc) return rc;
      DD_HIP(hipStreamSynchronize(s));
      for (int ek = 0; ek < NUM_EK; ++ek) {
        std::vector<uint8_t> packed;
        pack_conv_weights(wt.data(), conv_pack_geom2(23 - l, ek), ek, true, packed);
        rc = upload(h, L.wpackT[ek], packed.data(), packed.size(), s); if (rc) return rc;
        DD_HIP(hipStreamSynchronize(s));
      }
    }
    const std::vector<float>& gg = h->host_w[std::string(gn_names[l]) + ".weight"];
    const std::vector<float>& gb = h->host_w[std::string(gn_names[l]) + ".bias"];
    rc = upload(h, L.gamma, gg.data(), gg.size() * 4, s); if (rc) return rc;
    rc = upload(h, L.beta, gb.data(), gb.size() * 4, s); if (rc) return rc;
    DD_HIP(hipStreamSynchronize(s));
  }
  if (do_model && h->variant == DD_VARIANT_SWIN) {
    const char* names[2] = {"model.upsample_fuse.convA.conv", "model.upsample_fuse.convB.conv"};
    ConvLayer* Ls[2] = {&h->LA, &h->LB};
    for (int i = 0; i < 2; ++i) {
      ConvLayer& L = *Ls[i];
      L.cin = COND_C; L.cout = COND_C;
      const std::vector<float>& w = h->host_w[std::string(names[i]) + ".weight"];
      const std::vector<float>& b = h->host_w[std::string(names[i]) + ".bias"];
      for (int wi = 0; wi < NUM_WIMG; ++wi) {
        if (!wimg_has(wi, 5 + i)) continue;
        std::vector<uint8_t> packed;
        if (!pack_conv_weights(w.data(), conv_pack_geom2(5 + i, wimg_kind(wi)), wimg_kind(wi), true, packed)) split_fits = false;
        int rc = upload(h, L.wpack2[wi], packed.data(), packed.size(), s);
        if (rc) return rc;
        DD_HIP(hipStreamSynchronize(s));
      }
      int rc = upload(h, L.bias, b.data(), b.size() * 4, s); if (rc) return rc;
      rc = upload(h, L.w_oihw, w.data(), w.size() * 4, s); if (rc) return rc;       // fp32 OIHW: the hoisted form's E[t] tables (swin_ttab)
      DD_HIP(hipStreamSynchronize(s));
      // backward: data gradient of a 256->256 conv = the convB kernel (layer 6: raw input, no norm) on W^T flipped
      std::vector<float> wt(w.size());
      for (int co = 0; co < COND_C; ++co)
        for (int ci = 0; ci < COND_C; ++ci)
          for (int k = 0; k < 9; ++k) wt[((size_t)ci * COND_C + co) * 9 + (8 - k)] = w[((size_t)co * COND_C + ci) * 9 + k];
      for (int ek = 0; ek < NUM_EK; ++ek) {
        std::vector<uint8_t> packed;
        pack_conv_weights(wt.data(), conv_pack_geom2(6, ek), ek, true, packed);
        rc = upload(h, L.wpackT[ek], packed.data(), packed.size(), s); if (rc) return rc;
        DD_HIP(hipStreamSynchronize(s));
      }
    }
  }
  if (do_model) {
    h->split_ok = split_fits;      // a weight of magnitude >= 234 does not fit the split-f16 images: only the split modes refuse such parameters (check_split)
    const std::vector<float>& e = h->host_w["model.time_embedding.weight"];
    int rc = upload(h, h->emb, e.data(), e.size() * 4, s); if (rc) return rc;
    DD_HIP(hipStreamSynchronize(s));
    if (h->etab.bytes == 0) {
      DD_HIP(h->etab.alloc((size_t)EMB_ROWS * 10 * HID_C * 4));
      DD_HIP(h->zero_bias.alloc(COND_C * 4));
      DD_HIP(hipMemsetAsync(h->zero_bias.p, 0, COND_C * 4, s));
    }
    DD_HIP(launch_etab(h->L[2].w_oihw.as<float>(), h->emb.as<float>(), h->etab.as<float>(), s));
    DD_HIP(hipStreamSynchronize(s));
    h->committed = true;
  }
  if (do_fpn) {
    // ---- condition FPN: fold eval-mode BatchNorm into the (bias-free) convolutions, pack for the v2 kernels ----
    auto fold = [&](const std::string& pre, std::vector<double>& scale, std::vector<float>& shift) {
      const auto &g = h->host_w[pre + ".weight"], &b = h->host_w[pre + ".bias"], &m = h->host_w[pre + ".running_mean"],
                 &v = h->host_w[pre + ".running_var"];
      scale.resize(COND_C); shift.resize(COND_C);
      for (int c = 0; c < COND_C; ++c) {
        scale[c] = (double)g[c] / std::sqrt((double)v[c] + (double)BN_EPS);
        shift[c] = (float)((double)b[c] - (double)m[c] * scale[c]);
      }
    };
    std::vector<double> sc; std::vector<float> sh;
    bool fpn_fits = true;
    for (int i = 0; i < FPN_LEVELS; ++i) {
      const std::string pre = "conv_lateral." + std::to_string(i);
      fold(pre + ".1", sc, sh);
      const std::vector<float>& w0 = h->host_w[pre + ".0.weight"];    // [256][cin][3][3]
      const size_t per = (size_t)fpn_cin(h->variant, h->fpn_pyramid)[i] * 9;
      const size_t per_pad = (size_t)fpn_cin_pad(h->variant, h->fpn_pyramid)[i] * 9;      // zero weights for the padding channels
      std::vector<float> w((size_t)COND_C * per_pad, 0.f);
      for (int co = 0; co < COND_C; ++co)
        for (size_t k = 0; k < per; ++k) w[co * per_pad + k] = (float)((double)w0[co * per + k] * sc[co]);
      for (int wi = 0; wi <= WIMG_SPLIT; ++wi) {        // fp32, bf16, f16 and the split-f16 image (the split / refined modes' pyramid)
        std::vector<uint8_t> packed;
        if (!pack_conv_weights(w.data(), conv_pack_geom2(fpn_lat_layer(h->variant, h->fpn_pyramid, i), wimg_kind(wi)), wimg_kind(wi), true, packed)) fpn_fits = false;
        int rc = upload(h, h->fpn_lat_w[i][wi], packed.data(), packed.size(), s); if (rc) return rc;
        DD_HIP(hipStreamSynchronize(s));
      }
      int rc = upload(h, h->fpn_lat_b[i], sh.data(), sh.size() * 4, s); if (rc) return rc;
      DD_HIP(hipStreamSynchronize(s));
    }
    for (int j = 0; j < FPN_LEVELS - 1; ++j) {
      const std::string pre = "conv_up." + std::to_string(j);
      fold(pre + ".1", sc, sh);
      const std::vector<float>& wt = h->host_w[pre + ".0.weight"];     // ConvTranspose2d weight [cin][cout][2][2]
      // as a 1x1 convolution with 4 x 256 outputs: row (dy*2+dx)*256 + co, column ci
      std::vector<float> w((size_t)4 * COND_C * COND_C);
      std::vector<float> b4((size_t)4 * COND_C);
      for (int par = 0; par < 4; ++par)
        for (int co = 0; co < COND_C; ++co) {
          b4[par * COND_C + co] = sh[co];
          for (int ci = 0; ci < COND_C; ++ci)
            w[((size_t)par * COND_C + co) * COND_C + ci] = (float)((double)wt[((size_t)ci * COND_C + co) * 4 + par] * sc[co]);
        }
      for (int wi = 0; wi <= WIMG_SPLIT; ++wi) {
        std::vector<uint8_t> packed;
        if (!pack_conv_weights(w.data(), conv_pack_geom2(14, wimg_kind(wi)), wimg_kind(wi), true, packed)) fpn_fits = false;
        int rc = upload(h, h->fpn_up_w[j][wi], packed.data(), packed.size(), s); if (rc) return rc;
        DD_HIP(hipStreamSynchronize(s));
      }
      int rc = upload(h, h->fpn_up_b[j], b4.data(), b4.size() * 4, s); if (rc) return rc;
      DD_HIP(hipStreamSynchronize(s));
    }
    h->fpn_split_ok = fpn_fits;      // a folded weight beyond the split image's range: the split / refined modes keep the fp32-operand kernels for the pyramid
    h->fpn_committed = true;
  }
  if (do_neck) {
    // ---- HAHI neck: fold eval-mode BatchNorm into the bias-free convolutions (scale into the weights, shift = bias), pack ----
    // The kernels' channel counts can exceed the reference's (MPViT level 1: 216 carried as 224): the folded weights are laid into
    // [cout_k][cin_k] with zeros in the padding; a fusion convolution reads the concatenation [lateral (Ck) | projection (512)] (level 0:
    // [projection | lateral]), so its reference input channel ci >= C of the lateral part's successor moves up by Ck - C.
    bool neck_fits = true;
    for (const NeckConv& c : neck_convs(h->fpn_pyramid)) {
      const auto &g = h->host_w[c.name + ".bn.weight"], &b = h->host_w[c.name + ".bn.bias"], &m = h->host_w[c.name + ".bn.running_mean"],
                 &v = h->host_w[c.name + ".bn.running_var"];
      const std::vector<float>& w0 = h->host_w[c.name + ".conv.weight"];
      const PackGeom pg = conv_pack_geom2(c.layer, EK_F32);
      const int kk = c.ks * c.ks, cin_k = pg.cin, cout_k = pg.cout;
      std::vector<float> w((size_t)cout_k * cin_k * kk, 0.f), sh((size_t)pg.cout_pad, 0.f);
      for (int co = 0; co < c.cout; ++co) {
        const double sc = (double)g[co] / std::sqrt((double)v[co] + (double)BN_EPS);
        sh[co] = (float)((double)b[co] - (double)m[co] * sc);
        for (int ci = 0; ci < c.cin; ++ci) {
          int cik = ci;
          if (c.kind == 2 && c.level > 0 && ci >= c.C) cik = ci - c.C + c.Ck;          // [lateral | projection]: the projection part starts at Ck
          for (int k = 0; k < kk; ++k)
            w[((size_t)co * cin_k + cik) * kk + k] = (float)((double)w0[((size_t)co * c.cin + ci) * kk + k] * sc);
        }
      }
      const int slot = c.layer - neck_base(h->fpn_pyramid);
      for (int wi = 0; wi <= WIMG_SPLIT; ++wi) {
        std::vector<uint8_t> packed;
        if (!pack_conv_weights(w.data(), conv_pack_geom2(c.layer, wimg_kind(wi)), wimg_kind(wi), true, packed)) neck_fits = false;
        int rc = upload(h, h->neck_w[slot][wi], packed.data(), packed.size(), s); if (rc) return rc;
        DD_HIP(hipStreamSynchronize(s));
      }
      int rc = upload(h, h->neck_b[slot], sh.data(), sh.size() * 4, s); if (rc) return rc;
      DD_HIP(hipStreamSynchronize(s));
    }
    h->neck_split_ok = neck_fits;
    h->neck_committed = true;
  }
  if (!do_codec) return DD_OK;
  // ---- codec: fold eval-mode BatchNorm into the convolutions (reference depth_transform.py:15-26) ----
  auto W = [&](const char* n) -> const std::vector<float>& { return h->host_w[n]; };
  auto bn_fold = [&](const std::string& p, std::vector<float>& scale, std::vector<float>& shift) {
    const auto &g = W((p + ".weight").c_str()), &b = W((p + ".bias").c_str()), &m = W((p + ".running_mean").c_str()),
               &v = W((p + ".running_var").c_str());
    scale.resize(16); shift.resize(16);
    for (int c = 0; c < 16; ++c) {
      const double sc = (double)g[c] / std::sqrt((double)v[c] + (double)BN_EPS);
      scale[c] = (float)sc;
      shift[c] = (float)((double)b[c] - (double)m[c] * sc);
    }
  };
  std::vector<float> blob;
  auto push = [&](const std::vector<float>& v) { size_t off = blob.size(); blob.insert(blob.end(), v.begin(), v.end());
                                                 while (blob.size() % 4) blob.push_back(0.f); return off; };
  std::vector<float> sc, sh;
  bn_fold("depth_transform.conv_transform.0.1", sc, sh);
  std::vector<float> e0 = W("depth_transform.conv_transform.0.0.weight");
  for (int c = 0; c < 16; ++c) for (int k = 0; k < 9; ++k) e0[c * 9 + k] *= sc[c];
  const size_t o_e0 = push(e0), o_eb0 = push(sh);
  bn_fold("depth_transform.conv_transform.1.1", sc, sh);
  std::vector<float> e1 = W("depth_transform.conv_transform.1.0.weight");
  for (int co = 0; co < 16; ++co) for (int k = 0; k < 16 * 9; ++k) e1[co * 144 + k] *= sc[co];
  const size_t o_e1 = push(e1), o_eb1 = push(sh);
  std::vector<float> e1t(2304);          // the same weights tap-major [tap][ci][co] for enc1_kernel
  for (int co = 0; co < 16; ++co) for (int ci = 0; ci < 16; ++ci) for (int k = 0; k < 9; ++k) e1t[(k * 16 + ci) * 16 + co] = e1[(co * 16 + ci) * 9 + k];
  const size_t o_e1t = push(e1t);
  bn_fold("depth_transform.conv_inv_transform.1", sc, sh);
  std::vector<float> d0 = W("depth_transform.conv_inv_transform.0.weight");       // (in, out, 4, 4)
  for (int ci = 0; ci < 16; ++ci) for (int co = 0; co < 16; ++co) for (int k = 0; k < 16; ++k) d0[(ci * 16 + co) * 16 + k] *= sc[co];
  std::vector<float> db0(16);
  { const auto& cb = W("depth_transform.conv_inv_transform.0.bias"); for (int c = 0; c < 16; ++c) db0[c] = cb[c] * sc[c] + sh[c]; }
  const size_t o_d0 = push(d0), o_db0 = push(db0);
  // the same weights tap-major [ky][kx][ci][co] for the fused decoder kernel (a wave reads one (tap, ci) row of 16 couts uniformly)
  std::vector<float> d0t(4096);
  for (int ci = 0; ci < 16; ++ci) for (int co = 0; co < 16; ++co) for (int k = 0; k < 16; ++k) d0t[(k * 16 + ci) * 16 + co] = d0[(ci * 16 + co) * 16 + k];
  const size_t o_d0t = push(d0t);
  const size_t o_d1 = push(W("depth_transform.conv_inv_transform.3.0.weight"));
  {
    int rc = upload(h, h->codec_buf, blob.data(), blob.size() * 4, s); if (rc) return rc;
    DD_HIP(hipStreamSynchronize(s));
  }
  const float* base = h->codec_buf.as<float>();
  h->codec.enc_w0 = base + o_e0; h->codec.enc_b0 = base + o_eb0;
  h->codec.enc_w1 = base + o_e1; h->codec.enc_b1 = base + o_eb1; h->codec.enc_w1t = base + o_e1t;
  h->codec.dec_w0 = base + o_d0; h->codec.dec_b0 = base + o_db0; h->codec.dec_w0t = base + o_d0t;
  h->codec.dec_w1 = base + o_d1;
  h->codec.dec_b1 = W("depth_transform.conv_inv_transform.3.0.bias")[0];
  h->codec_committed = true;
  return DD_OK;
}

int dd_set_schedule(dd_handle_t h, const float* alphas_cumprod, int num_train_timesteps) {
  if (!h) return DD_ERR_INVALID_ARG;
  if (!alphas_cumprod || num_train_timesteps <= 0) return h->fail(DD_ERR_INVALID_ARG, "dd_set_schedule: null table or non-positive length");
  if (num_train_timesteps > EMB_ROWS)
    return h->fail(DD_ERR_INVALID_ARG, "dd_set_schedule: num_train_timesteps exceeds the time embedding's " + std::to_string(EMB_ROWS) +
                                       " rows (reference ...res.py:313 nn.Embedding(1280, 256))");
  for (int i = 0; i < num_train_timesteps; ++i)
    if (!(alphas_cumprod[i] > 0.f && alphas_cumprod[i] <= 1.f))
      return h->fail(DD_ERR_INVALID_ARG, "dd_set_schedule: alphas_cumprod must lie in (0, 1]");
  DD_HIP(hipSetDevice(h->device));
  DD_HIP(hipDeviceSynchronize());
  h->plans.clear();                    // tables are baked into plans
  h->last_once_plan = nullptr;
  h->acp.assign(alphas_cumprod, alphas_cumprod + num_train_timesteps);
  h->n_train = num_train_timesteps;
  DD_HIP(h->d_acp.alloc((size_t)num_train_timesteps * 4));
  DD_HIP(hipMemcpy(h->d_acp.p, h->acp.data(), (size_t)num_train_timesteps * 4, hipMemcpyHostToDevice));
  return DD_OK;
}

int dd_set_option(dd_handle_t h, const char* key, int64_t value) {
  if (!h || !key) return DD_ERR_INVALID_ARG;
  const std::string k(key);
  if (k == "graph") h->use_graph = value != 0;
  else if (k == "timing") h->timing = value != 0;
  else if (k == "debug_sync") h->debug_sync = value != 0;
  else if (k == "ablate") {
    if (h->ablate != (int)value) {            // kernel parameters are baked into captured graphs
      DD_HIP(hipDeviceSynchronize());
      h->plans.clear();
      h->last_once_plan = nullptr;
    }
    h->ablate = (int)value;
  }
  else if (k == "swin_w5") {
    if (value < 0 || value > 1) return h->fail(DD_ERR_INVALID_ARG, "dd_set_option: swin_w5 must be 0 or 1");
    if (h->swin_w5 != (int)value) { DD_HIP(hipDeviceSynchronize()); h->plans.clear(); h->last_once_plan = nullptr; }      // buffers and graphs are laid out for it
    h->swin_w5 = (int)value;
  }
  else if (k == "hoist_cond") {
    if (value < -1 || value > 1) return h->fail(DD_ERR_INVALID_ARG, "dd_set_option: hoist_cond must be -1 (automatic), 0 or 1");
    h->hoist_cond = (int)value;
  }
  else if (k == "bf16_storage") {
    if (h->bf16_pure != (value != 0)) {       // buffers and graphs are laid out for the element kinds
      DD_HIP(hipDeviceSynchronize());
      h->plans.clear(); h->last_once_plan = nullptr;
      h->cond_bufs.clear(); h->fpn_cond.reset(); h->fpn_cond_key[3] = -1; h->fpn_out.reset(); h->fpn_work.reset();
    }
    h->bf16_pure = value != 0;
  }
  else if (k == "naive_wgrad") h->naive_wgrad = value != 0;
  else if (k == "keep_trajectory") h->keep_traj = value != 0;
  else if (k == "use_trajectory") h->use_traj = value;
  else if (k == "adjoint_tiled") h->adjoint_tiled = value != 0;
  else if (k == "streams") h->n_streams = value < 1 ? 1 : (int)value;
  else if (k == "keep_activations_mb") h->keep_act_mb = value < 0 ? 0 : value;
  else if (k == "big_tiles") {
    if (value < -1 || value > 1) return h->fail(DD_ERR_INVALID_ARG, "dd_set_option: big_tiles must be -1 (automatic), 0 or 1");
    if (h->big_tiles != (int)value) { DD_HIP(hipDeviceSynchronize()); h->plans.clear(); h->last_once_plan = nullptr; }      // tile shape is baked into buffers and graphs
    h->big_tiles = (int)value;
  }
  else if (k == "thin_slots") {
    if (value < 1 || value > 4096) return h->fail(DD_ERR_INVALID_ARG, "dd_set_option: thin_slots must be in [1, 4096]");
    if (h->thin_slots != (int)value) { DD_HIP(hipDeviceSynchronize()); h->plans.clear(); h->last_once_plan = nullptr; }
    h->thin_slots = (int)value;
  }
  else if (k == "one_buffer") {
    if (value < 0 || value > 2) return h->fail(DD_ERR_INVALID_ARG, "dd_set_option: one_buffer must be 0, 1 or 2");
    if (h->one_buffer != (int)value) { DD_HIP(hipDeviceSynchronize()); h->plans.clear(); h->last_once_plan = nullptr; }      // the kernel choice is baked into captured graphs
    h->one_buffer = (int)value;
  }
  else if (k == "cond_direct") {     // stage_condition runs eagerly in front of the loop's graph: nothing to invalidate
    if (value < 0 || value > 1) return h->fail(DD_ERR_INVALID_ARG, "dd_set_option: cond_direct must be 0 or 1");
    h->cond_direct = (int)value;
  }
  else if (k == "cond_split") {      // dd_condition / dd_neck_condition are eager (no graph holds their kernels): nothing to invalidate
    if (value < 0 || value > 1) return h->fail(DD_ERR_INVALID_ARG, "dd_set_option: cond_split must be 0 or 1");
    h->cond_split = (int)value;
  }
  else if (k == "thin_xcd") {
    if (value < 0 || value > 1) return h->fail(DD_ERR_INVALID_ARG, "dd_set_option: thin_xcd must be 0 or 1");
    if (h->thin_xcd != (int)value) { DD_HIP(hipDeviceSynchronize()); h->plans.clear(); h->last_once_plan = nullptr; }      // kernel parameters are baked into captured graphs
    h->thin_xcd = (int)value;
  }
  else if (k == "thin_stream") {
    if (value < 0 || value > 1) return h->fail(DD_ERR_INVALID_ARG, "dd_set_option: thin_stream must be 0 or 1");
    if (h->thin_stream != (int)value) {          // the kernel choice is baked into captured graphs
      DD_HIP(hipDeviceSynchronize());
      h->plans.clear(); h->last_once_plan = nullptr;
    }
    h->thin_stream = (int)value;
  }
  else if (k == "f16r_wide" || k == "f16r_p4" || k == "f16r_c1") {
    if (value < 0 || value > 1) return h->fail(DD_ERR_INVALID_ARG, "dd_set_option: " + k + " must be 0 or 1");
    int& opt = k == "f16r_wide" ? h->f16r_wide : k == "f16r_c1" ? h->f16r_c1 : h->f16r_p4;
    if (opt != (int)value) { DD_HIP(hipDeviceSynchronize()); h->plans.clear(); h->last_once_plan = nullptr; }      // buffers and graphs are laid out for it
    opt = (int)value;
  }
  else if (k == "phase_prof_buffer") h->prof_buf = reinterpret_cast<unsigned long long*>((uintptr_t)value);   // device pointer (0 = off)
  else if (k == "phase_prof_layer") h->prof_layer = (int)value;
  else if (k == "layer_timing") {
    drain_layer_events(h);
    h->layer_timing = value != 0;
    for (int i = 0; i < dd_handle_s::N_LAYER_SLOTS; ++i) { h->layer_ms[i] = 0; h->layer_cnt[i] = 0; }
  } else return h->fail(DD_ERR_INVALID_ARG, "dd_set_option: unknown key '" + k + "'");
  return DD_OK;
}

int dd_get_counter(dd_handle_t h, const char* key, int64_t* value) {
  if (!h || !key || !value) return DD_ERR_INVALID_ARG;
  const std::string k(key);
  if (k == "graph_launches") *value = h->n_graph_launches;
  else if (k == "eager_loops") *value = h->n_eager_loops;
  else if (k == "graph_capture_failures") *value = h->n_capture_failures;
  else if (k == "plans") *value = (int64_t)h->plans.size();
  else if (k == "neck_launches") *value = h->n_neck_launches;
  else if (k == "trajectory_ticket") *value = h->traj_serial;
  else if (k == "lane_calls") *value = h->n_lane_calls;
  else if (k == "cond_split_ok") *value = (h->fpn_committed && h->fpn_split_ok ? 1 : 0) | (h->neck_committed && h->neck_split_ok ? 2 : 0);   // bit 0: FPN, bit 1: neck weights fit the split-f16 images
  else if (k == "resident_slots") *value = h->resident_slots;      // workgroup slots at two per CU (2 x multiProcessorCount): what the tile rules compare tile counts with
  else if (k == "trajectory_reuses") *value = h->n_traj_reuse;
  else return h->fail(DD_ERR_INVALID_ARG, "dd_get_counter: unknown key '" + k + "'");
  return DD_OK;
}

int dd_get_layer_ms(dd_handle_t h, int layer, double* total_ms, int64_t* launches) {
  if (!h || layer < 1 || layer > dd_handle_s::N_LAYER_SLOTS || !total_ms || !launches) return DD_ERR_INVALID_ARG;
  drain_layer_events(h);
  *total_ms = h->layer_ms[layer - 1];
  *launches = h->layer_cnt[layer - 1];
  return DD_OK;
}

int dd_last_loop_ms(dd_handle_t h, float* ms) {
  if (!h || !ms) return DD_ERR_INVALID_ARG;
  *ms = 0.f;
  if (!h->ev_valid) return DD_OK;
  DD_HIP(hipEventSynchronize(h->ev1));
  DD_HIP(hipEventElapsedTime(ms, h->ev0, h->ev1));
  return DD_OK;
}

static int condition_impl(dd_handle_t h, const float* const* feats, const int* feat_h, const int* feat_w, int n_levels, int B,
                          float* cond_out, int precision, void* stream, bool with_neck);

int dd_condition(dd_handle_t h, const float* const* feats, const int* feat_h, const int* feat_w, int n_levels, int B,
                 float* cond_out, int precision, void* stream) {
  return condition_impl(h, feats, feat_h, feat_w, n_levels, B, cond_out, precision, stream, false);
}

int dd_neck_condition(dd_handle_t h, const float* const* feats, const int* feat_h, const int* feat_w, int n_levels, int B,
                      float* cond_out, int precision, void* stream) {
  if (!h) return DD_ERR_INVALID_ARG;
  if (!h->neck_committed) return h->fail(DD_ERR_STATE, "hahineck.* weights not committed (dd_set_weight for the lateral / projection / fusion convolutions, "
                                                       "dd_commit_weights; DD_VARIANT_SWIN)");
  return condition_impl(h, feats, feat_h, feat_w, n_levels, B, cond_out, precision, stream, true);
}

static int condition_impl(dd_handle_t h, const float* const* feats, const int* feat_h, const int* feat_w, int n_levels, int B,
                          float* cond_out, int precision, void* stream, bool with_neck) {
  if (!h) return DD_ERR_INVALID_ARG;
  if (!h->fpn_committed) return h->fail(DD_ERR_STATE, "conv_lateral.* / conv_up.* weights not committed (dd_set_weight, dd_commit_weights)");
  if (with_neck && h->variant != DD_VARIANT_SWIN)
    return h->fail(DD_ERR_UNSUPPORTED, "dd_neck_condition: the HAHI neck kernels are built for the Swin-L (192/384/768/1536) and MPViT-small (128/216/288/288) pyramids of DD_VARIANT_SWIN");
  if (n_levels != FPN_LEVELS || !feats || !feat_h || !feat_w) return h->fail(DD_ERR_INVALID_ARG, "dd_condition: expects 4 pyramid levels");
  if (precision < DD_PREC_FP32 || precision > DD_PREC_LAST) return h->fail(DD_ERR_INVALID_ARG, "dd_condition: precision must be fp32, bf16, f16, f16x3 or f16r");
  if (B <= 0) return h->fail(DD_ERR_INVALID_ARG, "dd_condition: B must be positive");
  for (int i = 0; i < FPN_LEVELS; ++i) {
    if (!feats[i] || feat_h[i] <= 0 || feat_w[i] <= 0) return h->fail(DD_ERR_INVALID_ARG, "dd_condition: null feature pointer or non-positive size");
    if ((long long)B * feat_h[i] * feat_w[i] * fpn_cin_pad(h->variant, h->fpn_pyramid)[i] >= (1LL << 31) * 4) return h->fail(DD_ERR_INVALID_ARG, "tensor too large");
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  DD_HIP(hipSetDevice(h->device));
  int ek = ek_of_precision(precision, h->bf16_pure);      // EK_BF16M: inner tensors bf16, the result (level-0 lateral conv) f16
  // split / refined f16: both modes read an fp32 condition map; the once-per-image pyramid runs on the split-f16 kernels (fp32 tensors, f16-pair
  // operands, three MFMAs per product: ~22 mantissa bits at five times the fp32-operand MFMA rate) when its folded weights fit their images,
  // else -- or with option "cond_split" = 0 -- on the fp32-operand kernels
  bool split_pyr = false;
  if (ek == EK_F16S || ek == EK_F16R) {
    split_pyr = h->cond_split && h->fpn_split_ok && (!with_neck || h->neck_split_ok);
    ek = EK_F32;                                           // tensor kind of the pyramid (layouts, conversions, workspace)
  }
  const int kk = split_pyr ? (int)EK_F16S : ek;            // kernel kind of its convolutions
  const int wk = split_pyr ? WIMG_SPLIT : opnd_kind(ek);   // their weight image
  const int ok = opnd_kind(ek), sk = store_kind(ek);
  const size_t es = ek_size(ek);
  // workspace for this pyramid shape
  FpnWork* fw = h->fpn_work.get();
  bool same = fw && fw->B == B && fw->ek == ek && fw->pyr == h->fpn_pyramid && (fw->neck || !with_neck);
  for (int i = 0; same && i < FPN_LEVELS; ++i) same = fw->hs[i] == feat_h[i] && fw->ws[i] == feat_w[i];
  if (!same) {
    DD_HIP(hipDeviceSynchronize());
    h->fpn_work.reset(new FpnWork());
    fw = h->fpn_work.get();
    fw->B = B; fw->ek = ek; fw->pyr = h->fpn_pyramid;
    for (int i = 0; i < FPN_LEVELS; ++i) {
      fw->hs[i] = feat_h[i]; fw->ws[i] = feat_w[i];
      const size_t px = (size_t)B * feat_h[i] * feat_w[i];
      DD_HIP(fw->fin[i].alloc(px * fpn_cin_pad(h->variant, h->fpn_pyramid)[i] * es));
      if (with_neck) {
        DD_HIP(fw->nk_cat[i].alloc(px * (neck_ck(h->fpn_pyramid, i) + 512) * es));
        DD_HIP(fw->nk_out[i].alloc(px * neck_ck(h->fpn_pyramid, i) * es));
        fw->neck = true;
      }
      if (i > 0) {
        DD_HIP(fw->lat[i].alloc(px * COND_C * es));
        DD_HIP(fw->up[i - 1].alloc(px * 4 * COND_C * es));
        if (2 * feat_h[i] != feat_h[i - 1] || 2 * feat_w[i] != feat_w[i - 1])
          DD_HIP(fw->pooled[i - 1].alloc((size_t)B * feat_h[i - 1] * feat_w[i - 1] * COND_C * es));
      }
    }
  }
  // Res: the result IS the plan's condition buffer (latent size).  Swin: the map stays at the pyramid's finest size here
  // and is bilinearly upsampled into the plan's buffer by dd_denoise (cond == NULL), which knows the latent size.
  std::shared_ptr<DevBuf> cbuf;
  int rc = DD_OK;
  const bool swin = h->variant == DD_VARIANT_SWIN;
  if (swin) {
    const size_t need = (size_t)B * feat_h[0] * feat_w[0] * COND_C * es;
    if (!h->fpn_out || h->fpn_out->bytes < need) { h->fpn_out = std::make_shared<DevBuf>(); DD_HIP(h->fpn_out->alloc(need)); }
    cbuf = h->fpn_out;
  } else {
    rc = get_cond_buf(h, B, feat_h[0], feat_w[0], precision, &cbuf);
    if (rc) return rc;
  }

  auto launch = [&](int layer, const ConvParams& q) -> hipError_t {
    if (!h->layer_timing) return launch_conv_igemm2(layer, kk, q, s);
    hipEvent_t a, b;
    hipError_t e = hipEventCreate(&a); if (e != hipSuccess) return e;
    e = hipEventCreate(&b); if (e != hipSuccess) return e;
    (void)hipEventRecord(a, s);
    e = launch_conv_igemm2(layer, kk, q, s);
    (void)hipEventRecord(b, s);
    h->pending_ev.emplace_back(layer - 1, a, b);
    return e;
  };
  // top-down pass (reference ...res.py:108-118): x_3 = lat_3(f_3);  x_i = lat_i(f_i) + pool(up_i(x_{i+1}))
  for (int i = FPN_LEVELS - 1; i >= 0; --i) {
    const int hh = feat_h[i], ww = feat_w[i];
    DD_HIP(launch_nchw_to_nhwc_padded(feats[i], fw->fin[i].p, ok, B, fpn_cin(h->variant, h->fpn_pyramid)[i],
                                      fpn_cin_pad(h->variant, h->fpn_pyramid)[i], hh, ww, 1, s));
    const int lat_layer = fpn_lat_layer(h->variant, h->fpn_pyramid, i);
    const void* lat_in = fw->fin[i].p;
    if (with_neck) {
      // HAHI neck of level i (reference hahi.py:170-173,196-197,226-272; attention off): l = lateral(x); e = proj(l);
      // out = fusion(cat) with cat = [e | l] at level 0 (hahi.py:249: cat([fusion_res_conv, feat_conv])) and [l | e] above (:262)
      const int C = neck_ck(h->fpn_pyramid, i), CT = C + 512, l_off = (i == 0) ? 512 : 0, e_off = (i == 0) ? 0 : C;
      const int nb = neck_base(h->fpn_pyramid);
      ConvParams q{};
      q.B = B; q.h = hh; q.w = ww;
      q.tiles_x = (ww + 31) / 32;
      q.tiles_y = (hh + 7) / 8;
      q.in = fw->fin[i].p; q.in_cstride = C; q.in_coff = 0;
      q.out = fw->nk_cat[i].p; q.out_cstride = CT; q.out_coff = l_off;
      q.wpack = h->neck_w[i][wk].p; q.bias = h->neck_b[i].as<float>();
      DD_HIP(launch(nb + i, q));
      q.in = fw->nk_cat[i].p; q.in_cstride = CT; q.in_coff = l_off;
      q.out = fw->nk_cat[i].p; q.out_cstride = CT; q.out_coff = e_off;
      q.wpack = h->neck_w[4 + i][wk].p; q.bias = h->neck_b[4 + i].as<float>();
      DD_HIP(launch(nb + 4 + i, q));
      q.in = fw->nk_cat[i].p; q.in_cstride = CT; q.in_coff = 0;
      q.out = fw->nk_out[i].p; q.out_cstride = C; q.out_coff = 0;
      q.wpack = h->neck_w[8 + i][wk].p; q.bias = h->neck_b[8 + i].as<float>();
      DD_HIP(launch(nb + 8 + i, q));
      h->n_neck_launches += 3;
      lat_in = fw->nk_out[i].p;
    }
    ConvParams p{};
    p.B = B; p.h = hh; p.w = ww;
    p.tiles_x = (ww + 31) / 32;
    p.tiles_y = (hh + conv_pack_geom2(lat_layer, kk).th - 1) / conv_pack_geom2(lat_layer, kk).th;
    p.in = lat_in; p.wpack = h->fpn_lat_w[i][wk].p; p.bias = h->fpn_lat_b[i].as<float>();
    p.out = (i == 0) ? cbuf->p : fw->lat[i].p;
    p.addend = (i == FPN_LEVELS - 1) ? nullptr : (fw->pooled[i].p ? fw->pooled[i].p : fw->up[i].p);
    DD_HIP(launch(lat_layer, p));
    if (i > 0) {
      ConvParams u{};
      u.B = B; u.h = hh; u.w = ww;
      u.tiles_x = (ww + 31) / 32;
      u.tiles_y = (hh + conv_pack_geom2(14, kk).th - 1) / conv_pack_geom2(14, kk).th;
      u.in = fw->lat[i].p; u.wpack = h->fpn_up_w[i - 1][wk].p; u.bias = h->fpn_up_b[i - 1].as<float>(); u.out = fw->up[i - 1].p;
      DD_HIP(launch(14, u));
      if (fw->pooled[i - 1].p)
        DD_HIP(launch_adaptive_pool_blocked(fw->up[i - 1].p, fw->pooled[i - 1].p, ok, B, COND_C, 2 * hh, 2 * ww, feat_h[i - 1], feat_w[i - 1], s));
    }
    if (h->debug_sync) DD_HIP(hipStreamSynchronize(s));
  }
  if (cond_out) DD_HIP(launch_blocked_to_nchw(cbuf->p, sk, cond_out, B, COND_C, feat_h[0], feat_w[0], 0, s));
  h->fpn_cond = cbuf;
  h->fpn_cond_key[0] = B; h->fpn_cond_key[1] = feat_h[0]; h->fpn_cond_key[2] = feat_w[0]; h->fpn_cond_key[3] = precision;
  return DD_OK;
}

}  // extern "C"

namespace {
// The loop on B images (a whole call, or one lane of it: images img0 .. img0 + B - 1 of a batch of whole_B whose tensors start at the
// pointers given -- already offset to the lane's first image) on stream s.
int denoise_lane(dd_handle_t h, const float* x_T, const float* cond, float* x_0, int B, int lat_h, int lat_w,
                 int cond_h, int cond_w, int T, int precision, hipStream_t s, int lane, int img0, int whole_B, int64_t ticket, int S) {
  int rc = DD_OK;
  const bool timed = h->timing && lane == 0 && B == whole_B;      // lanes: the caller brackets fork .. join
  Plan* pl = nullptr;
  int keep = (h->keep_traj && precision != DD_PREC_NAIVE_FP32) ? 1 : 0;
  if (keep) {
    // ... and every step's raw activations as well when they fit the budget (option "keep_activations_mb", default 64 GiB of the 288):
    // the backward then recomputes nothing.  KITTI, T = 20: 1.8 GB (Res) / 4.0 GB (Swin) per image
    const size_t es = ek_size(store_kind(ek_of_precision(precision, h->bf16_pure)));
    const size_t per_step = (size_t)B * lat_h * lat_w * ((2 * HID_C + COND_C + (h->variant == DD_VARIANT_SWIN ? 2 * COND_C : 0)) * es + LATENT_C * 4);
    const size_t need = per_step * (size_t)T;
    if (need <= ((size_t)h->keep_act_mb << 20) &&
        (h->plans.count(PlanKey{B, lat_h, lat_w, cond_h, cond_w, T, precision, want_hoist(h, precision, T, 2), 2, lane, S}) || keep2_fits(h, need))) keep = 2;
  }
  rc = get_plan(h, PlanKey{B, lat_h, lat_w, cond_h, cond_w, T, precision, want_hoist(h, precision, T, keep), keep, lane, S}, &pl);
  if (rc && keep == 2) {
    // the per-step slots did not fit after all (fragmentation, another process): states only -- the backward then recomputes the activations
    (void)hipGetLastError();
    h->plans.erase(PlanKey{B, lat_h, lat_w, cond_h, cond_w, T, precision, want_hoist(h, precision, T, 2), 2, lane, S});
    keep = 1;
    rc = get_plan(h, PlanKey{B, lat_h, lat_w, cond_h, cond_w, T, precision, want_hoist(h, precision, T, keep), keep, lane, S}, &pl);
  }
  if (rc) return rc;
  const size_t n16 = (size_t)B * lat_h * lat_w * LATENT_C;
  float* x_first = keep ? pl->xstash.as<float>() : pl->x[0].as<float>();                                        // state entering step 0
  const float* x_last = keep ? pl->xstash.as<float>() + (size_t)(T - 1) * n16 : pl->x[(T - 1) & 1].as<float>();  // state entering step T-1
  if (keep) pl->traj_ticket = 0;         // being overwritten

  DD_HIP(launch_nchw_to_nhwc(x_T, x_first, EK_F32, B, LATENT_C, lat_h, lat_w, 0, s));
  rc = stage_condition(h, pl, cond, B, lat_h, lat_w, cond_h, cond_w, precision, s, img0, whole_B);
  if (rc) return rc;
  if (pl->exec && pl->graph_cond != pl->cond_ptr()) {      // the graph holds another condition pointer (the whole batch's buffer moved)
    DD_HIP(hipStreamSynchronize(s));
    (void)hipGraphExecDestroy(pl->exec);
    pl->exec = nullptr;
  }

  if (timed) {
    if (!h->ev0) { DD_HIP(hipEventCreate(&h->ev0)); DD_HIP(hipEventCreate(&h->ev1)); }
    DD_HIP(hipEventRecord(h->ev0, s));
  }
  if (precision == DD_PREC_NAIVE_FP32) {
    DD_HIP(hipMemsetAsync(pl->stats.p, 0, pl->stats_bytes, s));
    for (int k = 0; k < T; ++k) {
      float* xc = pl->x[k & 1].as<float>();
      float* xn = pl->x[(k + 1) & 1].as<float>();
      rc = enqueue_naive_eps(h, pl, k, xc, pl->tsteps.as<long long>(), k, 0, s);
      if (rc) return rc;
      DD_HIP(launch_naive_axpby(xc, pl->eps.as<float>(), pl->c1c2.as<float>(), k, xn, (long long)B * lat_h * lat_w * LATENT_C, s));
    }
    h->n_eager_loops++;
    if (timed) { DD_HIP(hipEventRecord(h->ev1, s)); h->ev_valid = true; }
    DD_HIP(launch_nhwc_to_nchw_f32(pl->x[T & 1].p, EK_F32, x_0, B, LATENT_C, lat_h, lat_w, 0, s));
    return DD_OK;
  }

  const bool want_graph = h->use_graph && !h->layer_timing && !h->debug_sync && !pl->capture_failed;
  bool launched = false;
  if (want_graph) {
    if (!pl->exec) {
      // first use of this plan: one eager pass sets function attributes / loads code objects and
      // validates the launch configuration before we capture
      rc = enqueue_loop_body(h, pl, s);
      if (rc) return rc;
      DD_HIP(hipStreamSynchronize(s));
      hipGraph_t graph = nullptr;
      hipError_t e = hipSuccess;
      if (!h->cap_stream) e = hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking);
      if (e == hipSuccess) e = hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeThreadLocal);
      if (e == hipSuccess) {
        int brc = enqueue_loop_body(h, pl, h->cap_stream);
        hipError_t e2 = hipStreamEndCapture(h->cap_stream, &graph);
        if (brc == DD_OK && e2 == hipSuccess && graph) {
          e = hipGraphInstantiate(&pl->exec, graph, nullptr, nullptr, 0);
          if (e != hipSuccess) pl->exec = nullptr;
          pl->graph_cond = pl->cond_ptr();
        }
        if (graph) (void)hipGraphDestroy(graph);
      }
      if (!pl->exec) {
        (void)hipGetLastError();
        pl->capture_failed = true;
        h->n_capture_failures++;
      }
      // the eager pass above consumed x[0]: restore the input state before the real run
      DD_HIP(launch_nchw_to_nhwc(x_T, x_first, EK_F32, B, LATENT_C, lat_h, lat_w, 0, s));
      if (timed) DD_HIP(hipEventRecord(h->ev0, s));
    }
    if (pl->exec) {
      DD_HIP(hipGraphLaunch(pl->exec, s));
      h->n_graph_launches++;
      launched = true;
    }
  }
  if (!launched) {
    rc = enqueue_loop_body(h, pl, s);
    if (rc) return rc;
    h->n_eager_loops++;
  }
  if (timed) { DD_HIP(hipEventRecord(h->ev1, s)); h->ev_valid = true; }
  // x_0 = c1*x + c2*relu(gn4(y4)) of the last step, written NCHW
  DD_HIP(launch_final(x_last, static_cast<const float*>(pl->slot(pl->y4, T - 1)), pl->stat_ptr(T - 1, 3), h->L[3].gamma.as<float>(),
                      h->L[3].beta.as<float>(), pl->c1c2.as<float>(), T - 1, 0, x_0, B, lat_h, lat_w, s));
  if (keep) { pl->traj_ticket = ticket; pl->traj_weights = h->weights_serial; pl->traj_consumed = false; }       // one ticket per dd_denoise call, shared by its lanes
  if (h->debug_sync) DD_HIP(hipStreamSynchronize(s));
  return DD_OK;
}
}  // namespace

extern "C" {

int dd_denoise(dd_handle_t h, const float* x_T, const float* cond, float* x_0, int B, int lat_h, int lat_w,
               int cond_h, int cond_w, int T, int precision, void* stream) {
  int rc = check_common(h, B, lat_h, lat_w, cond_h, cond_w, true);
  if (rc) return rc;
  if (!x_T || !x_0) return h->fail(DD_ERR_INVALID_ARG, "dd_denoise: null tensor pointer");
  if (T <= 0 || T > h->n_train) return h->fail(DD_ERR_INVALID_ARG, "dd_denoise: num_inference_steps must be in [1, num_train_timesteps]");
  if (precision < DD_PREC_NAIVE_FP32 || precision > DD_PREC_LAST) return h->fail(DD_ERR_INVALID_ARG, "dd_denoise: unknown precision");
  rc = check_split(h, precision, "dd_denoise"); if (rc) return rc;
  if (h->variant == DD_VARIANT_SWIN && precision == DD_PREC_NAIVE_FP32)
    return h->fail(DD_ERR_UNSUPPORTED, "DD_VARIANT_SWIN runs on the fused kernels only (no naive path)");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  DD_HIP(hipSetDevice(h->device));
  // Option "streams" = S > 1: the images of a batch are independent (GroupNorm is per sample), so the batch runs as S concurrent
  // sub-batches, lane 0 on the caller's stream and the others on streams of the handle, forked and joined with events on the caller's
  // stream.  Each lane has its own plan (activation buffers, hipGraph); the workgroups of one lane's kernels fill the tail of the
  // other's and hide its kernel boundaries.  Not for the naive path or the per-launch timing mode.  A training forward
  // ("keep_trajectory") keeps every lane's states and activations in that lane's plan under ONE ticket; dd_denoise_backward splits alike.
  const int S = lane_count(h, B, precision);
  const int64_t ticket = (h->keep_traj && precision != DD_PREC_NAIVE_FP32) ? ++h->traj_serial : 0;
  if (h->variant == DD_VARIANT_SWIN && h->swin_w5 && want_hoist(h, precision, T, h->keep_traj ? 1 : 0)) { rc = ensure_swin_w5(h, s); if (rc) return rc; }
  if (S <= 1) return denoise_lane(h, x_T, cond, x_0, B, lat_h, lat_w, cond_h, cond_w, T, precision, s, 0, 0, B, ticket, 1);
  for (int l = 1; l < S; ++l) {
    if (!h->lane_stream[l]) DD_HIP(hipStreamCreateWithFlags(&h->lane_stream[l], hipStreamNonBlocking));
    if (!h->lane_done[l]) DD_HIP(hipEventCreateWithFlags(&h->lane_done[l], hipEventDisableTiming));
  }
  if (!h->lane_fork) DD_HIP(hipEventCreateWithFlags(&h->lane_fork, hipEventDisableTiming));
  if (h->timing) {
    if (!h->ev0) { DD_HIP(hipEventCreate(&h->ev0)); DD_HIP(hipEventCreate(&h->ev1)); }
    DD_HIP(hipEventRecord(h->ev0, s));
  }
  DD_HIP(hipEventRecord(h->lane_fork, s));
  const size_t n_x = (size_t)LATENT_C * lat_h * lat_w, n_c = (size_t)COND_C * cond_h * cond_w;
  int img0 = 0;
  for (int l = 0; l < S; ++l) {
    const int n = B / S + (l < B % S ? 1 : 0);
    hipStream_t ls = l == 0 ? s : h->lane_stream[l];
    if (l > 0) DD_HIP(hipStreamWaitEvent(ls, h->lane_fork, 0));
    rc = denoise_lane(h, x_T + img0 * n_x, cond ? cond + img0 * n_c : nullptr, x_0 + img0 * n_x, n, lat_h, lat_w, cond_h, cond_w, T, precision,
                      ls, l, img0, B, ticket, S);
    if (l > 0) {                                           // join even after an error: the caller's stream must not run ahead of a lane
      (void)hipEventRecord(h->lane_done[l], ls);
      (void)hipStreamWaitEvent(s, h->lane_done[l], 0);
    }
    if (rc) return rc;
    img0 += n;
  }
  h->n_lane_calls++;
  if (h->timing) { DD_HIP(hipEventRecord(h->ev1, s)); h->ev_valid = true; }
  return DD_OK;
}

int dd_denoise_trace(dd_handle_t h, const float* x_T, const float* cond, float* states, int B, int lat_h, int lat_w,
                     int cond_h, int cond_w, int T, int precision, void* stream) {
  int rc = check_common(h, B, lat_h, lat_w, cond_h, cond_w, true);
  if (rc) return rc;
  if (!x_T || !states) return h->fail(DD_ERR_INVALID_ARG, "dd_denoise_trace: null tensor pointer");
  if (T <= 0 || T > h->n_train) return h->fail(DD_ERR_INVALID_ARG, "dd_denoise_trace: num_inference_steps must be in [1, num_train_timesteps]");
  if (precision < DD_PREC_NAIVE_FP32 || precision > DD_PREC_LAST) return h->fail(DD_ERR_INVALID_ARG, "dd_denoise_trace: unknown precision");
  rc = check_split(h, precision, "dd_denoise_trace"); if (rc) return rc;
  if (h->variant == DD_VARIANT_SWIN && precision == DD_PREC_NAIVE_FP32)
    return h->fail(DD_ERR_UNSUPPORTED, "DD_VARIANT_SWIN runs on the fused kernels only (no naive path)");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  DD_HIP(hipSetDevice(h->device));
  Plan* pl = nullptr;
  rc = get_plan(h, PlanKey{B, lat_h, lat_w, cond_h, cond_w, T, precision, want_hoist(h, precision, T, 0)}, &pl);    // the plan of a ONE-lane dd_denoise call: the same kernels -- a multi-lane dd_denoise at KITTI size takes the 16x32-tile form of the hoisted conv3 (plan_big_tiles), whose GroupNorm partial sums add up in another order
  if (rc) return rc;
  const bool naive = precision == DD_PREC_NAIVE_FP32;
  const size_t n16 = (size_t)B * lat_h * lat_w * LATENT_C;
  // X[k] = state entering step k (the stash of the loop backward); X[T] only on the unfused path
  if (pl->xstash.bytes < (size_t)(T + 1) * n16 * 4) DD_HIP(pl->xstash.alloc((size_t)(T + 1) * n16 * 4));
  float* X = pl->xstash.as<float>();
  const long long* ts = pl->tsteps.as<long long>();
  DD_HIP(launch_nchw_to_nhwc(x_T, X, EK_F32, B, LATENT_C, lat_h, lat_w, 0, s));
  rc = stage_condition(h, pl, cond, B, lat_h, lat_w, cond_h, cond_w, precision, s);
  if (rc) return rc;
  DD_HIP(hipMemsetAsync(pl->stats.p, 0, pl->stats_bytes, s));
  for (int k = 0; k < T; ++k) {
    if (naive) {
      rc = enqueue_naive_eps(h, pl, k, X + (size_t)k * n16, ts, k, 0, s);
      if (rc) return rc;
      DD_HIP(launch_naive_axpby(X + (size_t)k * n16, pl->eps.as<float>(), pl->c1c2.as<float>(), k, X + (size_t)(k + 1) * n16, (long long)n16, s));
    } else {
      // conv1 of step k applies the update of step k-1 (reads X[k-1] and y4 of step k-1) and leaves x_k in X[k]
      rc = enqueue_fused_step(h, pl, k, (k == 0) ? X : X + (size_t)(k - 1) * n16, X + (size_t)k * n16, k > 0, ts, k, 0, s);
      if (rc) return rc;
    }
  }
  h->n_eager_loops++;
  // states[j] = sample after step j: X[j+1] for j < T-1; the last one is the final update (launch_final, as in dd_denoise)
  for (int j = 0; j + 1 < T; ++j)
    DD_HIP(launch_nhwc_to_nchw_f32(X + (size_t)(j + 1) * n16, EK_F32, states + (size_t)j * n16, B, LATENT_C, lat_h, lat_w, 0, s));
  if (naive) {
    DD_HIP(launch_nhwc_to_nchw_f32(X + (size_t)T * n16, EK_F32, states + (size_t)(T - 1) * n16, B, LATENT_C, lat_h, lat_w, 0, s));
  } else {
    DD_HIP(launch_final(X + (size_t)(T - 1) * n16, pl->y4.as<float>(), pl->stat_ptr(T - 1, 3), h->L[3].gamma.as<float>(),
                        h->L[3].beta.as<float>(), pl->c1c2.as<float>(), T - 1, 0, states + (size_t)(T - 1) * n16, B, lat_h, lat_w, s));
  }
  if (h->debug_sync) DD_HIP(hipStreamSynchronize(s));
  return DD_OK;
}

int dd_denoise_once(dd_handle_t h, const float* x_t, const int64_t* t, const float* cond, float* eps, int B, int lat_h,
                    int lat_w, int cond_h, int cond_w, int precision, void* stream) {
  int rc = check_common(h, B, lat_h, lat_w, cond_h, cond_w, false);
  if (rc) return rc;
  if (!x_t || !t || !eps) return h->fail(DD_ERR_INVALID_ARG, "dd_denoise_once: null pointer");
  if (precision < DD_PREC_NAIVE_FP32 || precision > DD_PREC_LAST) return h->fail(DD_ERR_INVALID_ARG, "dd_denoise_once: unknown precision");
  rc = check_split(h, precision, "dd_denoise_once"); if (rc) return rc;
  if (h->variant == DD_VARIANT_SWIN && precision == DD_PREC_NAIVE_FP32)
    return h->fail(DD_ERR_UNSUPPORTED, "DD_VARIANT_SWIN runs on the fused kernels only (no naive path)");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  DD_HIP(hipSetDevice(h->device));
  Plan* pl = nullptr;
  rc = get_plan(h, PlanKey{B, lat_h, lat_w, cond_h, cond_w, 0, precision, want_hoist(h, precision, 0, 0)}, &pl);
  if (rc) return rc;
  const long long* tv = reinterpret_cast<const long long*>(t);
  DD_HIP(launch_nchw_to_nhwc(x_t, pl->x[0].p, EK_F32, B, LATENT_C, lat_h, lat_w, 0, s));
  rc = stage_condition(h, pl, cond, B, lat_h, lat_w, cond_h, cond_w, precision, s);
  if (rc) return rc;
  DD_HIP(hipMemsetAsync(pl->stats.p, 0, pl->stats_bytes, s));
  if (precision == DD_PREC_NAIVE_FP32) {
    rc = enqueue_naive_eps(h, pl, 0, pl->x[0].as<float>(), tv, 0, 1, s);
    if (rc) return rc;
    DD_HIP(launch_nhwc_to_nchw_f32(pl->eps.p, EK_F32, eps, B, LATENT_C, lat_h, lat_w, 0, s));
  } else {
    rc = enqueue_fused_step(h, pl, 0, pl->x[0].as<float>(), pl->x[1].as<float>(), false, tv, 0, 1, s);
    if (rc) return rc;
    DD_HIP(launch_final(pl->x[0].as<float>(), pl->y4.as<float>(), pl->stat_ptr(0, 3), h->L[3].gamma.as<float>(),
                        h->L[3].beta.as<float>(), pl->c1c2.as<float>(), 0, 1, eps, B, lat_h, lat_w, s));
  }
  h->last_once_plan = pl;
  // "keep_trajectory": the raw conv outputs and GroupNorm sums this call leaves in the plan are what dd_denoise_once_backward would
  // recompute; a ticket lets it read them instead (same validity rules as for the loop)
  pl->traj_ticket = (h->keep_traj && precision != DD_PREC_NAIVE_FP32) ? ++h->traj_serial : 0;
  pl->traj_weights = h->weights_serial;
  if (h->debug_sync) DD_HIP(hipStreamSynchronize(s));
  return DD_OK;
}

namespace {

float* grad_buf(dd_handle_t h, const std::string& name, size_t numel, hipStream_t s, hipError_t* err, int lane = 0) {
  auto& set = h->grads[lane];
  auto it = set.find(name);
  if (it == set.end()) {
    std::unique_ptr<DevBuf> b(new DevBuf());
    *err = b->alloc(numel * 4);
    if (*err != hipSuccess) return nullptr;
    *err = hipMemsetAsync(b->p, 0, numel * 4, s);
    it = set.emplace(name, std::move(b)).first;
  }
  return it->second->as<float>();
}

}  // namespace

int dd_zero_grad(dd_handle_t h, void* stream) {
  if (!h) return DD_ERR_INVALID_ARG;
  DD_HIP(hipSetDevice(h->device));
  for (int l = 0; l < dd_handle_s::MAX_LANES; ++l)      // set 0 = what dd_get_grad reads; sets 1.. = lane scratch (zero unless a call failed midway)
    for (auto& kv : h->grads[l]) DD_HIP(hipMemsetAsync(kv.second->p, 0, kv.second->bytes, reinterpret_cast<hipStream_t>(stream)));
  return DD_OK;
}

int dd_get_grad(dd_handle_t h, const char* name, float* dst, int64_t numel, void* stream) {
  if (!h || !name || !dst) return h ? h->fail(DD_ERR_INVALID_ARG, "dd_get_grad: null argument") : DD_ERR_INVALID_ARG;
  auto it = h->grads[0].find(name);
  if (it == h->grads[0].end()) return h->fail(DD_ERR_STATE, std::string("dd_get_grad: no gradient accumulated for '") + name + "'");
  if ((size_t)numel * 4 != it->second->bytes)
    return h->fail(DD_ERR_INVALID_ARG, std::string("dd_get_grad: '") + name + "' has " + std::to_string(it->second->bytes / 4) + " elements");
  DD_HIP(hipSetDevice(h->device));
  DD_HIP(hipMemcpyAsync(dst, it->second->p, it->second->bytes, hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(stream)));
  return DD_OK;
}

namespace {

int ensure_bwd_buffers(dd_handle_t h, Plan* pl) {
  if (pl->gA.p) return DD_OK;
  const size_t px = (size_t)pl->key.B * pl->key.h * pl->key.w;
  const size_t es = ek_size(pl->ek);
  DD_HIP(pl->gA.alloc(px * COND_C * 4));
  DD_HIP(pl->gY.alloc(px * COND_C * 4));
  DD_HIP(pl->dgb.alloc((size_t)pl->key.B * COND_C * 4 * sizeof(double)));
  if (pl->key.prec != DD_PREC_NAIVE_FP32) {
    DD_HIP(pl->bX.alloc(px * LATENT_C * es));
    DD_HIP(pl->bA1.alloc(px * HID_C * es));
    DD_HIP(pl->bF.alloc(px * COND_C * es));
    DD_HIP(pl->bA3.alloc(px * HID_C * es));
  }
  return DD_OK;
}

// Backward of one epsilon-network evaluation at state x (fp32 NHWC, device), timesteps tv[t_base + b * t_bstride], with the
// condition map already staged in the plan.  On entry pl->gA holds dLoss/deps (fp32 NHWC16); on exit it holds dLoss/dx
// (fp32 NHWC16).  Recomputes the forward pass (GroupNorm sums in stat slot 0), accumulates the parameter gradients into
// h->grads, writes (or accumulates) dLoss/dcond as NCHW fp32 into grad_cond when that is not NULL.
int bwd_core(dd_handle_t h, Plan* pl, const float* x_nhwc, const long long* tv, int t_base, int t_bstride, float* grad_cond,
             int accumulate_cond, hipStream_t s, const Plan* kept = nullptr, int kstep = 0, int lane = 0) {
  // kept != NULL: the forward pass of this step is NOT recomputed -- its raw conv outputs and GroupNorm sums are read from slot `kstep`
  // of the forward plan that kept them (PlanKey::keep == 2; same shape and element kinds as `pl`)
  const Plan* src = kept ? kept : pl;
  const int sstep = kept ? kstep : 0;
  auto st = [&](int layer) { return src->stat_ptr(sstep, layer); };
  const int B = pl->key.B, lat_h = pl->key.h, lat_w = pl->key.w, precision = pl->key.prec;
  const bool naive = precision == DD_PREC_NAIVE_FP32;
  const long long HW = (long long)lat_h * lat_w;
  const int mode = pl->ek;                  // kernels of the recomputed forward pass and of the data gradients
  const int ek = opnd_kind(mode);           // gradients, materialised activations, weight-gradient operands: bf16 in the mode EK_BF16M
  const int yk = store_kind(mode);          // the stored conv outputs y1..y3 and the condition map (f16 in that mode)
  int rc = DD_OK;
  if (!kept) DD_HIP(hipMemsetAsync(pl->stat_ptr(0, 0), 0, (size_t)4 * B * STAT_SLOTS * STAT_STRIDE * sizeof(double), s));
  const int lay = naive ? 0 : 1;                    // activation layout flag of the views: plain NHWC fp32 / channel-blocked
  const ActView nothing{nullptr, EK_F32, 0, 1, HW};
  if (naive) {
    rc = enqueue_naive_eps(h, pl, 0, x_nhwc, tv, t_base, t_bstride, s);
    if (rc) return rc;
  } else {
    if (!kept) rc = enqueue_fused_step(h, pl, 0, x_nhwc, pl->x[1].as<float>(), false, tv, t_base, t_bstride, s);
    if (rc) return rc;
    // the convs' input activations in the kernels' own element kind: x, a1 = relu(gn1(y1)), f = relu(gn2(y2)) + cond + E[t],
    // a3 = relu(gn3(y3))  (the weight gradients contract the conv-output gradients with these)
    DD_HIP(launch_view_copy(ActView{x_nhwc, EK_F32, 0, LATENT_C, HW}, ActView{pl->bX.p, ek, 0, LATENT_C, HW}, B, s));
    const void* ys[3] = {src->slot(src->y1, sstep), src->slot(src->y2, sstep), src->slot(src->y3, sstep)};
    const DevBuf* as[3] = {&pl->bA1, &pl->bF, &pl->bA3};
    for (int l = 0; l < 3; ++l) {
      const int C = kCouts[l];
      if (ek != EK_F32) {
        DD_HIP(launch_gn_bwd_apply_blocked(nullptr, ys[l], ek, yk, st(l), h->L[l].gamma.as<float>(), h->L[l].beta.as<float>(),
                                           nullptr, nullptr, as[l]->p, (l == 1) ? pl->cond_ptr() : nullptr, h->emb.as<float>(), tv, t_base,
                                           t_bstride, B, C, HW, s));
        continue;
      }
      const ActView yv{ys[l], ek, 1, C, HW}, av{as[l]->p, ek, 1, C, HW};
      const ActView cv = (l == 1) ? ActView{pl->cond_ptr(), ek, 1, C, HW} : nothing;
      DD_HIP(launch_gn_bwd_apply(nothing, yv, st(l), h->L[l].gamma.as<float>(), h->L[l].beta.as<float>(), pl->dgb.as<double>(),
                                 nothing, av, cv, h->emb.as<float>(), tv, t_base, t_bstride, B, s));
    }
  }
  // ---- backward, last layer first ----
  const void* ybuf[4] = {src->slot(src->y1, sstep), src->slot(src->y2, sstep), src->slot(src->y3, sstep), src->slot(src->y4, sstep)};
  const void* sa_buf = src->slot(src->sa, sstep);
  const void* sf_buf = src->slot(src->sf, sstep);
  const bool swin = h->variant == DD_VARIANT_SWIN;     // fused modes only (checked by the callers)
  // the conv's input activation; Swin: pred.0 reads the raw convB output sf (and convB reads sa, convA reads bF = u)
  const void* inbuf[4] = {naive ? (const void*)x_nhwc : pl->bX.p, naive ? pl->a1.p : pl->bA1.p,
                          naive ? pl->f.p : (swin ? sf_buf : pl->bF.p), naive ? pl->a3.p : pl->bA3.p};
  hipError_t e = hipSuccess;
  for (int l = 3; l >= 0; --l) {
    const int C = kCouts[l], CI = kCins[l];
    // conv4's output y4 and the incoming grad_eps are fp32 NHWC in every mode; everything else is in the plan's element kind
    const int ek_y = (naive || l == 3) ? EK_F32 : ek, ek_g = naive ? EK_F32 : ek;
    const ActView yv{ybuf[l], ek_y, lay, C, HW}, gav{pl->gA.p, (l == 3) ? (int)EK_F32 : ek_g, lay, C, HW}, gyv{pl->gY.p, ek_g, lay, C, HW};
    const float* gamma = h->L[l].gamma.as<float>();
    const float* beta = h->L[l].beta.as<float>();
    float* dgam = grad_buf(h, std::string(kGnNames[l]) + ".weight", C, s, &e, lane); DD_HIP(e);
    float* dbet = grad_buf(h, std::string(kGnNames[l]) + ".bias", C, s, &e, lane); DD_HIP(e);
    float* dbias = grad_buf(h, std::string(kConvNames[l]) + ".bias", C, s, &e, lane); DD_HIP(e);
    // 16-bit channel-blocked tensors (layers 0..2 of the fused bf16 / f16 modes): vectorised kernels, one pass for all sums
    const bool vec = !naive && ek != EK_F32 && l < 3;
    if (vec) {
      DD_HIP(hipMemsetAsync(pl->dgb.p, 0, (size_t)B * C * 4 * sizeof(double), s));
      DD_HIP(launch_gn_bwd_reduce_blocked(pl->gA.p, ybuf[l], ek, yk, st(l), gamma, beta, pl->dgb.as<double>(), B, C, HW, s));
      DD_HIP(launch_gn_bwd_apply_blocked(pl->gA.p, ybuf[l], ek, yk, st(l), gamma, beta, pl->dgb.as<double>(), pl->gY.p, nullptr,
                                         nullptr, nullptr, nullptr, 0, 0, B, C, HW, s));
      float* demb = nullptr;
      if (l == 1) { demb = grad_buf(h, "model.time_embedding.weight", (size_t)EMB_ROWS * COND_C, s, &e, lane); DD_HIP(e); }
      DD_HIP(launch_gn_param_grad4(pl->dgb.as<double>(), st(l), gamma, dgam, dbet, dbias, demb, tv, t_base, t_bstride, B, C, HW, s));
    } else {
      DD_HIP(hipMemsetAsync(pl->dgb.p, 0, (size_t)B * C * 2 * sizeof(double), s));
      DD_HIP(launch_gn_bwd_reduce(gav, yv, st(l), gamma, beta, pl->dgb.as<double>(), B, s));
      DD_HIP(launch_gn_bwd_apply(gav, yv, st(l), gamma, beta, pl->dgb.as<double>(), gyv, nothing, nothing, nullptr, nullptr, 0, 0, B, s));
      DD_HIP(launch_gn_param_grad(pl->dgb.as<double>(), dgam, dbet, B, C, s));
      DD_HIP(launch_channel_sum(gyv, dbias, nullptr, 0, 0, B, s));
    }
    float* dw = grad_buf(h, std::string(kConvNames[l]) + ".weight", (size_t)C * CI * 9, s, &e, lane); DD_HIP(e);
    const ActView inv{inbuf[l], ek_g, lay, CI, HW};
    if (!naive && ek != EK_F32 && !h->naive_wgrad) {
      const size_t need = wgrad_workspace_bytes(C, CI, B, lat_h, lat_w);
      if (h->wgrad_ws[lane].bytes < need) { DD_HIP(hipStreamSynchronize(s)); DD_HIP(h->wgrad_ws[lane].alloc(need)); }
      DD_HIP(launch_wgrad_mfma(pl->gY.p, inbuf[l], dw, h->wgrad_ws[lane].as<float>(), ek, C, CI, B, lat_h, lat_w, s, h->active_lanes));
    } else {
      DD_HIP(launch_naive_wgrad(gyv, inv, dw, B, lat_h, lat_w, s));     // fp32 operands: the unfused kernel (parity modes)
    }
    // dgrad: g_in = conv3x3(g_y, W^T flipped): C -> CI channels
    if (naive) {
      DD_HIP(launch_naive_conv3x3(pl->gY.as<float>(), h->L[l].wT_oihw.as<float>(), nullptr, pl->gA.as<float>(), B, lat_h, lat_w, C, CI, s));
    } else {
      const int layer = 23 - l;
      ConvParams q{};
      q.B = B; q.h = lat_h; q.w = lat_w;
      q.tiles_x = (lat_w + 31) / 32;
      q.tiles_y = (lat_h + conv_pack_geom2(layer, ek).th - 1) / conv_pack_geom2(layer, ek).th;
      q.in = pl->gY.p; q.wpack = h->L[l].wpackT[ek].p; q.bias = h->zero_bias.as<float>(); q.out = pl->gA.p;
      DD_HIP(launch_conv_igemm2(layer, ek, q, s));          // data gradients: plain kinds (bf16 in the mode EK_BF16M)
    }
    if (l == 2 && swin) {
      // Swin fuse (reference ...swin_addHAHI.py:321-333,378): sf = convB(sa), sa = convA(u), u = relu(gn2(y2)) + up(cond) + E[t];
      // no norm / activation in between.  gA = dLoss/dsf on entry, dLoss/du on exit (gY is the scratch in between).
      const char* fuse[2] = {"model.upsample_fuse.convB.conv", "model.upsample_fuse.convA.conv"};
      ConvLayer* FL[2] = {&h->LB, &h->LA};
      const void* fin[2] = {sa_buf, pl->bF.p};           // convB's input, convA's input
      void* gbuf[3] = {pl->gA.p, pl->gY.p, pl->gA.p};      // gradient w.r.t. sf -> sa -> u
      for (int i = 0; i < 2; ++i) {
        const ActView gout{gbuf[i], ek, 1, COND_C, HW}, fv{fin[i], ek, 1, COND_C, HW};
        float* db = grad_buf(h, std::string(fuse[i]) + ".bias", COND_C, s, &e, lane); DD_HIP(e);
        float* dwf = grad_buf(h, std::string(fuse[i]) + ".weight", (size_t)COND_C * COND_C * 9, s, &e, lane); DD_HIP(e);
        if (ek != EK_F32) DD_HIP(launch_channel_sum_blocked(gbuf[i], ek, db, B, COND_C, HW, s));
        else DD_HIP(launch_channel_sum(gout, db, nullptr, 0, 0, B, s));
        if (ek != EK_F32 && !h->naive_wgrad) {
          const size_t need = wgrad_workspace_bytes(COND_C, COND_C, B, lat_h, lat_w);
          if (h->wgrad_ws[lane].bytes < need) { DD_HIP(hipStreamSynchronize(s)); DD_HIP(h->wgrad_ws[lane].alloc(need)); }
          DD_HIP(launch_wgrad_mfma(gbuf[i], fin[i], dwf, h->wgrad_ws[lane].as<float>(), ek, COND_C, COND_C, B, lat_h, lat_w, s, h->active_lanes));
        } else {
          DD_HIP(launch_naive_wgrad(gout, fv, dwf, B, lat_h, lat_w, s));
        }
        ConvParams q{};
        q.B = B; q.h = lat_h; q.w = lat_w;
        q.tiles_x = (lat_w + 31) / 32;
        q.tiles_y = (lat_h + conv_pack_geom2(6, ek).th - 1) / conv_pack_geom2(6, ek).th;
        q.in = gbuf[i]; q.wpack = FL[i]->wpackT[ek].p; q.bias = h->zero_bias.as<float>(); q.out = gbuf[i + 1];
        DD_HIP(launch_conv_igemm2(6, ek, q, s));
      }
    }
    if (l == 2) {
      // gA = dLoss/df, f = relu(gn2(y2)) + cond + E[t]  (reference ...res.py:330-340): the same gradient reaches cond, E[t] and a2
      // (Swin: gA = dLoss/du and cond enters through the bilinear upsample, whose adjoint maps the gradient back to (ch, cw))
      const ActView gf{pl->gA.p, ek_g, lay, COND_C, HW};
      if (grad_cond && swin) {
        DD_HIP(launch_upsample_adjoint(pl->gA.p, ek, grad_cond, B, COND_C, pl->key.ch, pl->key.cw, lat_h, lat_w, accumulate_cond, s, h->adjoint_tiled));
      } else if (grad_cond) {
        if (!naive && ek != EK_F32) DD_HIP(launch_blocked_to_nchw(pl->gA.p, ek, grad_cond, B, COND_C, lat_h, lat_w, accumulate_cond, s));
        else DD_HIP(launch_view_to_nchw(gf, grad_cond, B, accumulate_cond, s));
      }
      if (naive || ek == EK_F32) {     // (16-bit modes: the layer-1 reduction pass below also sums g_f per channel)
        float* demb = grad_buf(h, "model.time_embedding.weight", (size_t)EMB_ROWS * COND_C, s, &e, lane); DD_HIP(e);
        DD_HIP(launch_channel_sum(gf, demb, tv, t_base, t_bstride, B, s));
      }
    }
    if (h->debug_sync) DD_HIP(hipStreamSynchronize(s));
  }
  return DD_OK;
}

int check_bwd(dd_handle_t h, int precision, const char* who) {
  if (h->variant == DD_VARIANT_SWIN && precision == DD_PREC_NAIVE_FP32)
    return h->fail(DD_ERR_UNSUPPORTED, std::string(who) + ": DD_VARIANT_SWIN has no unfused path (use fp32 / bf16 / f16)");
  if (precision < DD_PREC_NAIVE_FP32 || precision > DD_PREC_LAST) return h->fail(DD_ERR_INVALID_ARG, std::string(who) + ": unknown precision");
  if (precision == DD_PREC_F16X3 || precision == DD_PREC_F16R)
    return h->fail(DD_ERR_UNSUPPORTED, std::string(who) + ": DD_PREC_F16X3 (split f16) and DD_PREC_F16R (refined f16) are forward-only modes; train in fp32 / bf16 / f16");
  return DD_OK;
}

}  // namespace

int dd_denoise_once_backward(dd_handle_t h, const float* x_t, const int64_t* t, const float* cond, const float* grad_eps,
                             float* grad_x, float* grad_cond, int B, int lat_h, int lat_w, int cond_h, int cond_w,
                             int precision, void* stream) {
  int rc = check_common(h, B, lat_h, lat_w, cond_h, cond_w, false);
  if (rc) return rc;
  if (!x_t || !t || !grad_eps) return h->fail(DD_ERR_INVALID_ARG, "dd_denoise_once_backward: null pointer");
  rc = check_bwd(h, precision, "dd_denoise_once_backward");
  if (rc) return rc;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  DD_HIP(hipSetDevice(h->device));
  Plan* pl = nullptr;
  // the same kernels (hoisted or not) as the forward call: a recompute differentiates the function the forward evaluated
  rc = get_plan(h, PlanKey{B, lat_h, lat_w, cond_h, cond_w, 0, precision, want_hoist(h, precision, 0, 0)}, &pl);
  if (rc) return rc;
  rc = ensure_bwd_buffers(h, pl);
  if (rc) return rc;
  DD_HIP(launch_nchw_to_nhwc(x_t, pl->x[0].p, EK_F32, B, LATENT_C, lat_h, lat_w, 0, s));
  rc = stage_condition(h, pl, cond, B, lat_h, lat_w, cond_h, cond_w, precision, s);
  if (rc) return rc;
  DD_HIP(launch_nchw_to_nhwc(grad_eps, pl->gA.p, EK_F32, B, LATENT_C, lat_h, lat_w, 0, s));
  // the forward call's activations, if it kept them for us (option "use_trajectory" = its ticket): no recompute
  const int64_t ticket = h->use_traj;
  h->use_traj = 0;
  const Plan* kept = nullptr;
  if (ticket != 0 && precision != DD_PREC_NAIVE_FP32) {
    auto it = h->plans.find(PlanKey{B, lat_h, lat_w, cond_h, cond_w, 0, precision, want_hoist(h, precision, 0, 0)});
    if (it != h->plans.end() && it->second->traj_ticket == ticket && it->second->traj_weights == h->weights_serial && it->second->ek == pl->ek)
      kept = it->second.get();
  }
  if (kept) h->n_traj_reuse++;
  rc = bwd_core(h, pl, pl->x[0].as<float>(), reinterpret_cast<const long long*>(t), 0, 1, grad_cond, 0, s, kept, 0);
  if (rc) return rc;
  if (grad_x) DD_HIP(launch_nhwc_to_nchw_f32(pl->gA.p, EK_F32, grad_x, B, LATENT_C, lat_h, lat_w, 0, s));
  h->last_once_plan = pl;
  return DD_OK;
}

}  // extern "C"

namespace {
// The loop backward on B images (a whole call or one lane of it, see denoise_lane); parameter gradients go to gradient set `lane`.
int denoise_backward_lane(dd_handle_t h, const float* x_T, const float* cond, const float* grad_x0, float* grad_xT, float* grad_cond,
                          int B, int lat_h, int lat_w, int cond_h, int cond_w, int T, int precision, hipStream_t s, int lane, int img0,
                          int whole_B, int64_t ticket, bool* reused, int S) {
  int rc = DD_OK;
  Plan* pl = nullptr;
  rc = get_plan(h, PlanKey{B, lat_h, lat_w, cond_h, cond_w, T, precision, want_hoist(h, precision, T, 1), 0, lane, S}, &pl);      // recompute = the kernels of the forward that kept the trajectory
  if (rc) return rc;
  rc = ensure_bwd_buffers(h, pl);
  if (rc) return rc;
  const bool naive = precision == DD_PREC_NAIVE_FP32;
  const size_t n16 = (size_t)B * lat_h * lat_w * LATENT_C;
  // The states entering each step: kept by the forward call (option "keep_trajectory" + the ticket passed through "use_trajectory": same
  // shape, same parameters, nothing run on that plan since), else regenerated here by running the forward loop again.
  const Plan* kept = nullptr;
  for (int lvl = 2; lvl >= 1 && !kept && ticket != 0 && !naive; --lvl) {
    auto it = h->plans.find(PlanKey{B, lat_h, lat_w, cond_h, cond_w, T, precision, want_hoist(h, precision, T, lvl), lvl, lane, S});
    if (it != h->plans.end() && it->second->traj_ticket == ticket && it->second->traj_weights == h->weights_serial) kept = it->second.get();
  }
  const Plan* kept_act = (kept && kept->key.keep == 2 && kept->ek == pl->ek) ? kept : nullptr;      // activations too: no recompute
  *reused = kept != nullptr;
  // this backward consumes the ticket: from here on the plan's kept activations may be dropped when another shape needs the room
  // (keep2_fits; until then a second backward on the same ticket still finds them); the kernels enqueued below read them -- an eviction
  // synchronises the device first
  if (kept) const_cast<Plan*>(kept)->traj_consumed = true;
  const size_t need = (size_t)(kept ? 1 : T + 1) * n16 * 4;
  if (pl->xstash.bytes < need) DD_HIP(pl->xstash.alloc(need));
  float* Xown = pl->xstash.as<float>();
  const float* X = kept ? kept->xstash.as<float>() : Xown;      // X[k] = state entering step k (k < T)
  float* G = kept ? Xown : Xown + (size_t)T * n16;              // running dLoss/dx
  const long long* ts = pl->tsteps.as<long long>();
  rc = stage_condition(h, pl, cond, B, lat_h, lat_w, cond_h, cond_w, precision, s, img0, whole_B);
  if (rc) return rc;
  // ---- forward loop again, keeping every intermediate state (16 channels: T x 6.8 MB per KITTI image) ----
  if (!kept) DD_HIP(launch_nchw_to_nhwc(x_T, Xown, EK_F32, B, LATENT_C, lat_h, lat_w, 0, s));
  if (!kept) DD_HIP(hipMemsetAsync(pl->stats.p, 0, pl->stats_bytes, s));
  for (int k = 0; !kept && k + 1 < T; ++k) {         // the last step's epsilon is recomputed by its backward pass
    if (naive) {
      rc = enqueue_naive_eps(h, pl, k, Xown + (size_t)k * n16, ts, k, 0, s);
      if (rc) return rc;
      DD_HIP(launch_naive_axpby(Xown + (size_t)k * n16, pl->eps.as<float>(), pl->c1c2.as<float>(), k, Xown + (size_t)(k + 1) * n16, (long long)n16, s));
    } else {
      // conv1 of step k applies the update of step k-1 (reads X[k-1], y4 of step k-1) and writes X[k]
      rc = enqueue_fused_step(h, pl, k, (k == 0) ? Xown : Xown + (size_t)(k - 1) * n16, Xown + (size_t)k * n16, k > 0, ts, k, 0, s);
      if (rc) return rc;
    }
  }
  if (!kept && !naive && T > 1) {
    // X[T-1] = update of step T-2 applied to X[T-2]: the fused path does that inside the NEXT step's conv1
    rc = enqueue_fused_step(h, pl, T - 1, Xown + (size_t)(T - 2) * n16, Xown + (size_t)(T - 1) * n16, true, ts, T - 1, 0, s);
    if (rc) return rc;
  }
  // ---- backward through the chain x_{k+1} = c1_k x_k + c2_k eps(x_k, t_k, cond) ----
  DD_HIP(launch_nchw_to_nhwc(grad_x0, G, EK_F32, B, LATENT_C, lat_h, lat_w, 0, s));
  for (int k = T - 1; k >= 0; --k) {
    DD_HIP(launch_bwd_chain(G, pl->gA.as<float>(), pl->c1c2.as<float>(), k, 0, (long long)n16, s));          // gA = c2_k G
    rc = bwd_core(h, pl, X + (size_t)k * n16, ts, k, 0, grad_cond, k < T - 1 ? 1 : 0, s, kept_act, k, lane);
    if (rc) return rc;
    DD_HIP(launch_bwd_chain(G, pl->gA.as<float>(), pl->c1c2.as<float>(), k, 1, (long long)n16, s));          // G = c1_k G + gA
  }
  if (grad_xT) DD_HIP(launch_nhwc_to_nchw_f32(G, EK_F32, grad_xT, B, LATENT_C, lat_h, lat_w, 0, s));
  return DD_OK;
}
}  // namespace

extern "C" {

int dd_denoise_backward(dd_handle_t h, const float* x_T, const float* cond, const float* grad_x0, float* grad_xT, float* grad_cond,
                        int B, int lat_h, int lat_w, int cond_h, int cond_w, int T, int precision, void* stream) {
  int rc = check_common(h, B, lat_h, lat_w, cond_h, cond_w, true);
  if (rc) return rc;
  if (!x_T || !grad_x0) return h->fail(DD_ERR_INVALID_ARG, "dd_denoise_backward: null pointer");
  if (T <= 0 || T > h->n_train) return h->fail(DD_ERR_INVALID_ARG, "dd_denoise_backward: num_inference_steps must be in [1, num_train_timesteps]");
  rc = check_bwd(h, precision, "dd_denoise_backward");
  if (rc) return rc;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  DD_HIP(hipSetDevice(h->device));
  const int64_t ticket = h->use_traj;
  h->use_traj = 0;
  // Option "streams": the images' backward passes are independent except for the parameter gradients -- every lane accumulates into its
  // own gradient set and weight-gradient workspace, the sets of lanes 1.. are added into set 0 (what dd_get_grad reads) after the join.
  const int S = lane_count(h, B, precision);
  bool reused = false;
  if (S <= 1) {
    rc = denoise_backward_lane(h, x_T, cond, grad_x0, grad_xT, grad_cond, B, lat_h, lat_w, cond_h, cond_w, T, precision, s, 0, 0, B, ticket, &reused, 1);
    if (rc == DD_OK && reused) h->n_traj_reuse++;
    return rc;
  }
  for (int l = 1; l < S; ++l) {
    if (!h->lane_stream[l]) DD_HIP(hipStreamCreateWithFlags(&h->lane_stream[l], hipStreamNonBlocking));
    if (!h->lane_done[l]) DD_HIP(hipEventCreateWithFlags(&h->lane_done[l], hipEventDisableTiming));
  }
  if (!h->lane_fork) DD_HIP(hipEventCreateWithFlags(&h->lane_fork, hipEventDisableTiming));
  DD_HIP(hipEventRecord(h->lane_fork, s));
  const size_t n_x = (size_t)LATENT_C * lat_h * lat_w, n_c = (size_t)COND_C * cond_h * cond_w;
  int img0 = 0;
  bool all_reused = true;
  h->active_lanes = S;
  for (int l = 0; l < S; ++l) {
    const int n = B / S + (l < B % S ? 1 : 0);
    hipStream_t ls = l == 0 ? s : h->lane_stream[l];
    if (l > 0) DD_HIP(hipStreamWaitEvent(ls, h->lane_fork, 0));
    rc = denoise_backward_lane(h, x_T + img0 * n_x, cond ? cond + img0 * n_c : nullptr, grad_x0 + img0 * n_x, grad_xT ? grad_xT + img0 * n_x : nullptr,
                               grad_cond ? grad_cond + img0 * n_c : nullptr, n, lat_h, lat_w, cond_h, cond_w, T, precision, ls, l, img0, B, ticket, &reused, S);
    all_reused = all_reused && reused;
    if (l > 0) {
      (void)hipEventRecord(h->lane_done[l], ls);
      (void)hipStreamWaitEvent(s, h->lane_done[l], 0);
    }
    if (rc) {
      // a failed lane: the partial parameter gradients of the lane sets must not leak into the next call's sums
      h->active_lanes = 1;
      for (int q = 1; q < dd_handle_s::MAX_LANES; ++q)
        for (auto& kv : h->grads[q]) (void)hipMemsetAsync(kv.second->p, 0, kv.second->bytes, s);
      return rc;
    }
    img0 += n;
  }
  h->active_lanes = 1;
  // after the join, on the caller's stream: set 0 += set l, set l = 0
  for (int l = 1; l < S; ++l)
    for (auto& kv : h->grads[l]) {
      hipError_t e = hipSuccess;
      float* dst = grad_buf(h, kv.first, kv.second->bytes / 4, s, &e, 0); DD_HIP(e);
      DD_HIP(launch_add_inplace(dst, kv.second->as<float>(), (long long)(kv.second->bytes / 4), s));
      DD_HIP(hipMemsetAsync(kv.second->p, 0, kv.second->bytes, s));
    }
  if (all_reused) h->n_traj_reuse++;
  h->n_lane_calls++;
  return DD_OK;
}

int dd_add_noise(dd_handle_t h, const float* x0, const float* noise, const int64_t* t, float* out, int B, int C,
                 int lat_h, int lat_w, void* stream) {
  if (!h) return DD_ERR_INVALID_ARG;
  if (h->n_train <= 0) return h->fail(DD_ERR_STATE, "schedule not set (dd_set_schedule)");
  if (!x0 || !noise || !t || !out || B <= 0 || C <= 0 || lat_h <= 0 || lat_w <= 0)
    return h->fail(DD_ERR_INVALID_ARG, "dd_add_noise: null pointer or non-positive size");
  DD_HIP(hipSetDevice(h->device));
  DD_HIP(launch_add_noise(x0, noise, reinterpret_cast<const long long*>(t), h->d_acp.as<float>(), h->n_train, out, B,
                          (long long)C * lat_h * lat_w, reinterpret_cast<hipStream_t>(stream)));
  return DD_OK;
}

static int ensure_codec_tmp(dd_handle_t h, size_t bytes) {
  if (h->codec_tmp.bytes >= bytes) return DD_OK;
  DD_HIP(hipDeviceSynchronize());
  DD_HIP(h->codec_tmp.alloc(bytes));
  return DD_OK;
}

int dd_encode(dd_handle_t h, const float* depth, float* latent, int B, int H, int W, void* stream) {
  if (!h) return DD_ERR_INVALID_ARG;
  if (!h->codec_committed) return h->fail(DD_ERR_STATE, "depth_transform.* weights not committed");
  if (!depth || !latent || B <= 0 || H <= 0 || W <= 0) return h->fail(DD_ERR_INVALID_ARG, "dd_encode: null pointer or non-positive size");
  DD_HIP(hipSetDevice(h->device));
  const int lh = (H - 1) / 2 + 1, lw = (W - 1) / 2 + 1;
  int rc = ensure_codec_tmp(h, (size_t)B * lh * lw * LATENT_C * 4);
  if (rc) return rc;
  DD_HIP(launch_encode(h->codec, depth, h->codec_tmp.as<float>(), latent, B, H, W, reinterpret_cast<hipStream_t>(stream)));
  return DD_OK;
}

int dd_decode(dd_handle_t h, const float* latent, float* depth, int B, int lat_h, int lat_w, void* stream) {
  if (!h) return DD_ERR_INVALID_ARG;
  if (!h->codec_committed) return h->fail(DD_ERR_STATE, "depth_transform.* weights not committed");
  if (!latent || !depth || B <= 0 || lat_h <= 0 || lat_w <= 0) return h->fail(DD_ERR_INVALID_ARG, "dd_decode: null pointer or non-positive size");
  DD_HIP(hipSetDevice(h->device));
  int rc = ensure_codec_tmp(h, (size_t)B * (2 * lat_h) * (2 * lat_w) * LATENT_C * 4);
  if (rc) return rc;
  DD_HIP(launch_decode(h->codec, latent, h->codec_tmp.as<float>(), depth, B, lat_h, lat_w, reinterpret_cast<hipStream_t>(stream)));
  return DD_OK;
}

int dd_debug_fetch(dd_handle_t h, const char* name, float* out, int64_t numel, void* stream) {
  if (!h || !name || !out) return DD_ERR_INVALID_ARG;
  Plan* pl = h->last_once_plan;
  if (!pl) return h->fail(DD_ERR_STATE, "dd_debug_fetch: no dd_denoise_once call to inspect");
  const std::string n(name);
  const void* src = nullptr; int C = 0; int ek = store_kind(pl->ek);
  if (n == "y1") { src = pl->y1.p; C = HID_C; }
  else if (n == "y2") { src = pl->y2.p; C = COND_C; }
  else if (n == "y3") {
    src = pl->y3.p; C = HID_C;
    if (pl->ek == EK_F16R && pl->wide) return h->fail(DD_ERR_UNSUPPORTED, "dd_debug_fetch: y3 travels as scaled int16 in this plan (option f16r_wide = 1)");
  }
  else if (n == "y4") { src = pl->y4.p; C = LATENT_C; ek = EK_F32; }
  else return h->fail(DD_ERR_INVALID_ARG, "dd_debug_fetch: unknown tensor '" + n + "'");
  if (numel != (int64_t)pl->key.B * C * pl->key.h * pl->key.w) return h->fail(DD_ERR_INVALID_ARG, "dd_debug_fetch: numel mismatch");
  DD_HIP(launch_nhwc_to_nchw_f32(src, ek, out, pl->key.B, C, pl->key.h, pl->key.w, pl->key.prec != DD_PREC_NAIVE_FP32,
                                 reinterpret_cast<hipStream_t>(stream)));
  return DD_OK;
}

// FNV-1a over every packed / staged denoiser weight buffer as it sits in HBM (buffer sizes included): two handles fed the same
// parameters through different routes (dd_set_weight vs dd_set_weight_device) must agree.
int dd_debug_weights_digest(dd_handle_t h, uint64_t* digest) {
  if (!h || !digest) return DD_ERR_INVALID_ARG;
  if (!h->committed) return h->fail(DD_ERR_STATE, "dd_debug_weights_digest: model.* weights not committed");
  DD_HIP(hipSetDevice(h->device));
  DD_HIP(hipDeviceSynchronize());
  uint64_t d = 1469598103934665603ull;
  std::vector<uint8_t> tmp;
  auto eat = [&](const DevBuf& b) -> hipError_t {
    uint64_t n = b.bytes;
    for (int i = 0; i < 8; ++i) { d ^= (n >> (8 * i)) & 0xFF; d *= 1099511628211ull; }
    if (!b.p || !b.bytes) return hipSuccess;
    tmp.resize(b.bytes);
    hipError_t e = hipMemcpy(tmp.data(), b.p, b.bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) return e;
    for (uint8_t c : tmp) { d ^= c; d *= 1099511628211ull; }
    return hipSuccess;
  };
  auto eat_layer = [&](const ConvLayer& L) -> hipError_t {
    for (int ek = 0; ek < NUM_EK; ++ek) {
      hipError_t e = eat(L.wpack2[ek]); if (e != hipSuccess) return e;
      e = eat(L.wpackT[ek]); if (e != hipSuccess) return e;
    }
    { hipError_t e = eat(L.wpack2[WIMG_SPLIT]); if (e != hipSuccess) return e; }
    { hipError_t e = eat(L.wpack2[WIMG_STACK]); if (e != hipSuccess) return e; }
    const DevBuf* rest[5] = {&L.bias, &L.w_oihw, &L.wT_oihw, &L.gamma, &L.beta};
    for (const DevBuf* b : rest) { hipError_t e = eat(*b); if (e != hipSuccess) return e; }
    return hipSuccess;
  };
  for (int l = 0; l < 4; ++l) DD_HIP(eat_layer(h->L[l]));
  if (h->variant == DD_VARIANT_SWIN) { DD_HIP(eat_layer(h->LA)); DD_HIP(eat_layer(h->LB)); }
  DD_HIP(eat(h->emb));
  DD_HIP(eat(h->etab));
  *digest = d;
  return DD_OK;
}

}  // extern "C"
