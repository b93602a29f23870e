// dd_api_plans.cpp -- per-shape plans and the launch sequences (see dd_api_internal.h for the split of the C ABI).
#include "dd_api_internal.h"

namespace ddapi {

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
  if (precision == DD_PREC_NAIVE_FP32 || h->layer_timing || h->debug_sync || h->prof_buf || h->check_finite) S = 1;      // (check_finite: its counters and its report are per call, not per lane)
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
int get_cond_buf(dd_handle_t h, int B, int lh, int lw, int precision, std::shared_ptr<DevBuf>* out, int lane) {
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
int want_hoist(dd_handle_t h, int precision, int T, int keep) {
  if (precision == DD_PREC_NAIVE_FP32) return 0;
  const int ek = ek_of_precision(precision, h->bf16_pure);
  if (h->variant == DD_VARIANT_SWIN) {
    // (round 6) ONE refined-f16 call with per-sample timesteps (dd_denoise_once, T = 0: what a "fast" head's ddim_loss evaluates, ...swin_addHAHI.py:207-223)
    // runs the hoisted form as well -- its E[t] border tables are then built per IMAGE from the call's own timesteps (ConvParams::ttab_bstride);
    // every other precision keeps the reference's order for single calls
    if (T <= 0 && keep == 0 && ek == EK_F16R && h->swin_w5 && h->hoist_cond != 0) return 1;
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
  // Round 6: the Res denoiser keeps its 8x32 tiles -- in their one-patch-buffer form, three workgroups per CU (conv3h_kid) -- under concurrent lanes as
  // well.  Round 3 had picked the 16x32 tiles there (+1.8 % on that round's kernels); on every box since round 5's call 26 the one-buffer form is ahead
  // under two lanes: 556.7 vs 527.1 and 542.4 vs 522.4 maps/s at KITTI B = 4 f16r (profiles/r06_experiments.md section 2).  The Swin 5x5 form has no
  // one-buffer kernel and keeps the 16x32 tiles when its 8x32 tiles exceed the resident slots.
  const bool many = (long long)key.B * ((key.h + 7) / 8) * ((key.w + 31) / 32) > h->resident_slots;
  return many && h->variant == DD_VARIANT_SWIN;
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
    return h->fail(DD_ERR_UNSUPPORTED, "DD_PREC_F16R runs the hoisted forward-only plans: for the Swin / MPViT denoiser those exist with option swin_w5 = 1 "
                                       "(dd_denoise / dd_denoise_trace / dd_denoise_once); training plans: DD_PREC_F16");
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
    const int T1 = key.T > 0 ? key.T : key.B;      // (a single call: one table per image, built from that image's timestep)
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
      if (layer == 4 && rf) { q.persist_slots = h->thin_slots; q.cadd_scale = static_cast<const float*>(pl->slot(pl->y3_scale, step)); return launch_conv4_stream(EK_F16, q, s, true, pl->wide, pl->p4); }
      if (layer == 4 && stream4) { q.persist_slots = h->thin_slots; return launch_conv4_stream(tk, q, s); }
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
    p.ttab_bstride = k.T > 0 ? 0 : SWIN_TT_ROWS * HID_C;
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
int enqueue_cond_conv(dd_handle_t h, Plan* pl, hipStream_t s, const float* nchw) {
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
  if (k.T > 0 && pl->ttab_weights != h->weights_serial) {      // (a single call's tables depend on the call's timesteps: dd_denoise_once builds them)
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
                    int precision, hipStream_t s, int img0, int whole_B) {
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

}  // namespace ddapi
