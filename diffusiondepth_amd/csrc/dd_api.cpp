// dd_api.cpp -- C ABI (include/ddepth.h) over the HIP kernels: handle, parameter packing, per-shape
// plans (scratch + schedule tables + captured hipGraph of the T-step loop) and launch sequencing.
#include "dd_api_internal.h"
#include <cstdlib>

namespace ddapi {
thread_local std::string g_create_error;
}  // namespace ddapi


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
  {
    // hipGraph replay is the DEFAULT only where it is safe.  With the HIP 7.0 runtime's graph fast path ("packet capture": pre-built AQL packets copied into
    // the hardware queue) long runs of replays next to eager launches on the same stream give WRONG results in windows of ~70 launches every few
    // thousand packets -- 800 back-to-back eval forwards: 4 of 4 processes wrong from forward ~307 on, 0 of 4 with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
    // (profiles/r06_experiments.md section 10; the same switch separates round 5's intermittent training NaN).  The runtime reads that variable once, when it
    // initialises: the Python package, bench.py and the tests export it = 0 before their first HIP call, and a handle created in a process WITHOUT it
    // enqueues its loops eagerly (option "graph" = 1 overrides) -- the same kernels in the same order, measured at the same rate (557 vs 557 maps/s).
    const char* pc = getenv("DEBUG_CLR_GRAPH_PACKET_CAPTURE");
    h->use_graph = pc && pc[0] == '0' && pc[1] == 0;
    h->graph_off_by_env = !h->use_graph;
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
  for (hipStream_t b : h->burnt_streams) (void)hipStreamDestroy(b);
  for (hipEvent_t e : h->probe_events) (void)hipEventDestroy(e);
  delete h;
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
  else if (k == "check_finite") h->check_finite = (int)value;
  else if (k == "train_graphs") h->train_graphs = value != 0;
  else if (k == "graph_fence") h->graph_fence = (int)value;
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
  else if (k == "lane_probe") h->lane_probe = value != 0;
  else if (k == "x3_grad_fp32") h->x3_grad_fp32 = value != 0;
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
  else if (k == "graph_default") *value = h->graph_off_by_env ? 0 : 1;       // 0 = this handle defaulted to eager loops: DEBUG_CLR_GRAPH_PACKET_CAPTURE was not "0" when it was created
  else if (k == "eager_loops") *value = h->n_eager_loops;
  else if (k == "graph_capture_failures") *value = h->n_capture_failures;
  else if (k == "plans") *value = (int64_t)h->plans.size();
  else if (k == "neck_launches") *value = h->n_neck_launches;
  else if (k == "trajectory_ticket") *value = h->traj_serial;
  else if (k == "lane_calls") *value = h->n_lane_calls;
  else if (k == "lane_probe_retries") *value = h->n_lane_probe_retries;
  else if (k == "lane_overlap") *value = h->lane_overlap_seen;
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

namespace ddapi {
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
  // the graph holds another condition pointer (the whole batch's buffer moved).  Only plans whose LOOP reads the condition map care: the hoisted
  // plans (Res: conv3(cond) runs once per image in stage_condition, in front of the graph; Swin: the whole step-invariant chain does) pass the
  // pointer to kernels that never dereference it -- a refined-f16 lane that alternates between dd_condition's buffer and a direct tensor used to
  // re-capture on every switch
  if (pl->exec && !pl->key.hoist && pl->graph_cond != pl->cond_ptr()) {
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

  // A plan that KEEPS its trajectory -- the forward of a training step -- is enqueued eagerly (option "train_graphs" = 1 restores the graph).  Round 5
  // found 16-bit training steps of a whole head producing non-finite values intermittently (3 of 8 processes within 40 iterations, always in iterations
  // 20..34; never with fp32 operands, never forward-only) and the one library switch that separates clean from failing runs is whether this forward is a
  // replayed hipGraph (0 of 12 processes fail with eager launches, 9 of 21 with the graph: profiles/r05_experiments.md section 4).  Round 6 localised it
  // OUTSIDE the library (profiles/r06_experiments.md section 5): a torch-free driver of these very calls never fails (0 of 64 processes), the library alone
  // from Python never fails (0 of 24), the head fails only while PyTorch's MIOpen convolutions AND batch-norm training kernels run between the calls
  // (either half taken away: 0 of 10), and not with a host synchronisation on either side of hipGraphLaunch (option "graph_fence") or with the HIP
  // runtime's graph fast path off (DEBUG_CLR_GRAPH_PACKET_CAPTURE=0: 0 of 8 against 6 of 8) -- the runtime's packet-captured graph launch next to those
  // kernels.  The eager form costs the 59-ms training step nothing measurable; the inference plans (hundreds of replays per bench run, parity-checked,
  // no MIOpen kernel in between) are not affected.
  const bool want_graph = h->use_graph && !h->layer_timing && !h->debug_sync && !pl->capture_failed && (!keep || h->train_graphs);
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
      if (h->graph_fence & 1) DD_HIP(hipStreamSynchronize(s));
      if (h->graph_fence & 4) {
        // the replay on a stream of the handle's own, ordered against the caller's stream by two events (what the lanes do for lane 1..)
        if (!h->lane_fork) DD_HIP(hipEventCreateWithFlags(&h->lane_fork, hipEventDisableTiming));
        if (!h->lane_done[0]) DD_HIP(hipEventCreateWithFlags(&h->lane_done[0], hipEventDisableTiming));
        DD_HIP(hipEventRecord(h->lane_fork, s));
        DD_HIP(hipStreamWaitEvent(h->cap_stream, h->lane_fork, 0));
        DD_HIP(hipGraphLaunch(pl->exec, h->cap_stream));
        DD_HIP(hipEventRecord(h->lane_done[0], h->cap_stream));
        DD_HIP(hipStreamWaitEvent(s, h->lane_done[0], 0));
      } else
      DD_HIP(hipGraphLaunch(pl->exec, s));
      if (h->graph_fence & 2) DD_HIP(hipStreamSynchronize(s));
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
}  // namespace ddapi

namespace ddapi {
// A lane's stream has to run CONCURRENTLY with the caller's.  The HIP runtime multiplexes the streams of a priority level onto GPU_MAX_HW_QUEUES
// (default 4) hardware queues and, once that many exist, hands a new stream the least-used one -- in a process that created other streams first
// (an eagerly initialised RCCL communicator: every rank of a data-parallel job; measured in round 6, profiles/r06_experiments.md section 9) that
// is the caller's own queue, and two lanes in one hardware queue do not overlap: the KITTI B = 4 step ran 23 % slower than with concurrent lanes
// (and slower than as ONE lane: the fork / join events serialise inside the queue).  So
// the candidate is PROBED -- one idle ~100-us wavefront on each stream, forked and joined by events, each writing its own start / end ticks of the
// device clock: the two intervals coincide when the streams are concurrent and follow each other when they share a queue -- and a candidate that failed is kept alive (the next one then lands on another queue)
// while another is tried, at most GPU_MAX_HW_QUEUES times.  Once per handle and lane (~0.5 ms); option "lane_probe" = 0 skips it; counters
// "lane_overlap" (1 / 0 / -1 = not probed) and "lane_probe_retries".
int acquire_lane_stream(dd_handle_t h, int lane, hipStream_t caller) {
  hipStream_t cand = nullptr;
  DD_HIP(hipStreamCreateWithFlags(&cand, hipStreamNonBlocking));
#ifndef DD_HOST_EMULATION
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  const bool capturing = hipStreamIsCapturing(caller, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone;
  if (h->lane_probe && !capturing) {
    // The probe's clock is the DEVICE's: each idle wavefront writes its own start / end ticks.  Not hipEventElapsedTime -- a timing-enabled event that
    // another stream waits on (the fork below) made the HIP runtime's graph fast path corrupt launches on the caller's stream some 16 000 packets
    // later (300 back-to-back eval forwards went wrong at forward ~234 in 20 of 30 processes; none of 12 with events created with
    // hipEventDisableTiming, none with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0: profiles/r06_experiments.md section 10).  Every event the library lets a
    // stream wait on is created with hipEventDisableTiming; these three stay alive with the handle (the configuration that was soaked).
    hipEvent_t ea = nullptr, ed = nullptr, ec = nullptr;
    long long* stamps = nullptr;
    DD_HIP(hipEventCreateWithFlags(&ea, hipEventDisableTiming)); DD_HIP(hipEventCreateWithFlags(&ed, hipEventDisableTiming)); DD_HIP(hipEventCreateWithFlags(&ec, hipEventDisableTiming));
    h->probe_events.push_back(ea); h->probe_events.push_back(ed); h->probe_events.push_back(ec);
    DD_HIP(hipMalloc(&stamps, 4 * sizeof(long long)));
    const long long ticks = 10000;       // 100 us of the 100-MHz clock
    int rc_ = DD_OK;
    for (int attempt = 0; attempt < 4; ++attempt) {
      hipError_t e = launch_spin(1, nullptr, caller);                                     // code object loaded, both queues awake
      if (e == hipSuccess) e = launch_spin(1, nullptr, cand);
      if (e == hipSuccess) e = hipStreamSynchronize(caller);
      if (e == hipSuccess) e = hipStreamSynchronize(cand);
      if (e == hipSuccess) e = hipEventRecord(ea, caller);
      if (e == hipSuccess) e = hipStreamWaitEvent(cand, ea, 0);
      if (e == hipSuccess) e = launch_spin(ticks, stamps, caller);
      if (e == hipSuccess) e = launch_spin(ticks, stamps + 2, cand);
      if (e == hipSuccess) e = hipEventRecord(ec, cand);
      if (e == hipSuccess) e = hipStreamWaitEvent(caller, ec, 0);
      if (e == hipSuccess) e = hipEventRecord(ed, caller);
      if (e == hipSuccess) e = hipEventSynchronize(ed);
      long long t[4] = {0, 0, 0, 0};
      if (e == hipSuccess) e = hipMemcpy(t, stamps, sizeof t, hipMemcpyDeviceToHost);
      if (e != hipSuccess) { rc_ = h->fail(DD_ERR_HIP, std::string("lane-overlap probe: ") + hipGetErrorString(e)); break; }
      const long long lo = t[0] > t[2] ? t[0] : t[2], hi = t[1] < t[3] ? t[1] : t[3];      // intersection of the two idle intervals
      const bool overlap = hi - lo > ticks / 2;
      h->lane_overlap_seen = overlap ? 1 : 0;
      if (overlap || attempt == 3) break;
      h->burnt_streams.push_back(cand);      // stays alive: its queue keeps its reference, the next candidate lands elsewhere
      h->n_lane_probe_retries++;
      cand = nullptr;
      e = hipStreamCreateWithFlags(&cand, hipStreamNonBlocking);
      if (e != hipSuccess) { rc_ = h->fail(DD_ERR_HIP, std::string("lane-overlap probe: ") + hipGetErrorString(e)); break; }
    }
    (void)hipFree(stamps);
    if (rc_) { if (cand) (void)hipStreamDestroy(cand); return rc_; }
  }
#else
  (void)caller;
#endif
  h->lane_stream[lane] = cand;
  return DD_OK;
}
}  // namespace ddapi

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
    if (!h->lane_stream[l]) { rc = acquire_lane_stream(h, l, s); if (rc) return rc; }
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
  if (h->variant == DD_VARIANT_SWIN && pl->key.hoist) {
    // the hoisted single call (refined f16): one E[t] border table per image, from this call's timesteps (read and clamped on the device)
    if (!h->LA.w_oihw.p || !h->LB.w_oihw.p || !h->L[2].w_oihw.p) return h->fail(DD_ERR_STATE, "hoisted Swin form: the fp32 weights of the fuse convolutions are not on the device");
    DD_HIP(launch_swin_ttab(h->LA.w_oihw.as<float>(), h->LB.w_oihw.as<float>(), h->L[2].w_oihw.as<float>(), h->emb.as<float>(), tv, B, lat_h, lat_w,
                            pl->tt_scratch.as<float>(), pl->ttab.as<float>(), s));
  }
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
