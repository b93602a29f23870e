// dd_api_train.cpp -- the training entry points: backward of one denoiser call and of the T-step loop (see dd_api_internal.h).
#include "dd_api_internal.h"

namespace ddapi {

float* grad_buf(dd_handle_t h, const std::string& name, size_t numel, hipStream_t s, hipError_t* err, int lane) {
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

}  // namespace ddapi

extern "C" {
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

}  // extern "C"
namespace ddapi {

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
             int accumulate_cond, hipStream_t s, const Plan* kept, int kstep, int lane) {
  // kept != NULL: the forward pass of this step is NOT recomputed -- its raw conv outputs and GroupNorm sums are read from slot `kstep`
  // of the forward plan that kept them (PlanKey::keep == 2; same shape and element kinds as `pl`)
  const Plan* src = kept ? kept : pl;
  const int sstep = kept ? kstep : 0;
  auto st = [&](int layer) { return src->stat_ptr(sstep, layer); };
  const int B = pl->key.B, lat_h = pl->key.h, lat_w = pl->key.w, precision = pl->key.prec;
  const bool naive = precision == DD_PREC_NAIVE_FP32;
  const long long HW = (long long)lat_h * lat_w;
  const int mode = pl->ek;                  // kernels of the recomputed forward pass and of the data gradients
  // gradients, materialised activations, weight-gradient operands: bf16 in the mode EK_BF16M.  Split f16 (DD_PREC_F16X3, round 6): the forward the
  // backward differentiates -- recomputed here or kept by the forward call -- runs the split kernels (fp32 tensors in the fp32 mode's layouts, every
  // product as three f16 MFMAs: the values the loss sees hold the 1e-3 absolute depth bound, the ReLU masks and GroupNorm statistics are the exact
  // forward's); the gradients BEHIND it travel as f16 through the f16 mode's kernels (MFMA data and weight gradients; the GroupNorm-backward kernels read
  // the fp32 y / condition map and write f16) -- or, option "x3_grad_fp32" = 1, as fp32 through the fp32 mode's (unfused weight gradient: ~60x slower,
  // the parity form)
  const int ek = mode == EK_F16S ? (h->x3_grad_fp32 ? (int)EK_F32 : (int)EK_F16) : opnd_kind(mode);
  const int yk = store_kind(mode);          // the stored conv outputs y1..y3 and the condition map (f16 in that mode)
  int rc = DD_OK;
  // option "check_finite" (debug): after a stage, count the non-finite elements of what it wrote; the first hit fails the call by name
  auto chk = [&](const char* what, int l, const void* ptr, long long n, int kind) -> int {
    if (!h->check_finite) return DD_OK;
    if (h->check_finite == 2) {      // asynchronous: one counter per stage, read by check_finite_report at the end of the call
      if (!h->chk_buf.p || h->chk_labels.size() >= 16383) return DD_OK;      // (calls with the option set run as one lane -- lane_count -- so a report is always open; entry 16383 is mode 1's counter)
      DD_HIP(launch_count_nonfinite(ptr, n, kind, h->chk_buf.as<unsigned>() + h->chk_labels.size(), s));
      h->chk_labels.push_back(std::string(what) + " of layer " + std::to_string(l) + " (step slot " + std::to_string(sstep) + ", lane " + std::to_string(lane) + ")");
      return DD_OK;
    }
    // (mode 1 counts into the LAST entry of the option's own buffer: h->wmax is the weight-range scratch of the device weight route)
    if (!h->chk_buf.p) DD_HIP(h->chk_buf.alloc(16384 * sizeof(unsigned)));
    unsigned* cnt = h->chk_buf.as<unsigned>() + 16383;
    DD_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned), s));
    DD_HIP(launch_count_nonfinite(ptr, n, kind, cnt, s));
    unsigned c = 0;
    DD_HIP(hipMemcpyAsync(&c, cnt, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    DD_HIP(hipStreamSynchronize(s));
    if (c) return h->fail(DD_ERR_HIP, std::string("check_finite: ") + what + " of layer " + std::to_string(l) + " (step slot " + std::to_string(sstep) + ", lane " +
                                      std::to_string(lane) + ") holds " + std::to_string(c) + " non-finite values of " + std::to_string(n));
    return DD_OK;
  };
  const int kk = (ek == EK_F32) ? 0 : (ek == EK_BF16 ? 1 : 2);
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
  if (swin && mode == EK_F16S && ek != EK_F32) {
    // split f16 with f16 gradients: the fuse convolutions' raw results (fp32 in this mode) as f16 operands of their weight gradients
    const size_t bytes = (size_t)B * COND_C * HW * 2;
    if (pl->bSa.bytes < bytes) { DD_HIP(pl->bSa.alloc(bytes)); DD_HIP(pl->bSf.alloc(bytes)); }
    DD_HIP(launch_view_copy(ActView{sa_buf, EK_F32, 1, COND_C, HW}, ActView{pl->bSa.p, ek, 1, COND_C, HW}, B, s));
    DD_HIP(launch_view_copy(ActView{sf_buf, EK_F32, 1, COND_C, HW}, ActView{pl->bSf.p, ek, 1, COND_C, HW}, B, s));
    sa_buf = pl->bSa.p; sf_buf = pl->bSf.p;
  }
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
    rc = chk("GroupNorm-backward sums (dgb)", l, pl->dgb.p, (long long)B * C * (vec ? 4 : 2), 3); if (rc) return rc;
    rc = chk("dLoss/dy (gY)", l, pl->gY.p, (long long)B * C * HW, naive ? 0 : kk); if (rc) return rc;
    rc = chk("the conv's input activation", l, inbuf[l], (long long)B * CI * HW, naive ? 0 : kk); if (rc) return rc;
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
    rc = chk("weight gradient (dw, accumulated)", l, dw, (long long)C * CI * 9, 0); if (rc) return rc;
    rc = chk("dLoss/d(input) out of the data-gradient convolution (gA)", l, pl->gA.p, (long long)B * CI * HW, (naive || l == 0) ? 0 : kk); if (rc) return rc;
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

// option "check_finite" = 2: counters of the running backward call
int check_finite_begin(dd_handle_t h, hipStream_t s) {
  if (h->check_finite != 2) return DD_OK;
  if (!h->chk_buf.p) DD_HIP(h->chk_buf.alloc(16384 * sizeof(unsigned)));
  DD_HIP(hipMemsetAsync(h->chk_buf.p, 0, 16384 * sizeof(unsigned), s));
  h->chk_labels.clear();
  return DD_OK;
}
int check_finite_report(dd_handle_t h, hipStream_t s) {
  if (h->check_finite != 2 || h->chk_labels.empty()) return DD_OK;
  std::vector<unsigned> c(h->chk_labels.size());
  DD_HIP(hipMemcpyAsync(c.data(), h->chk_buf.p, c.size() * sizeof(unsigned), hipMemcpyDeviceToHost, s));
  DD_HIP(hipStreamSynchronize(s));
  for (size_t i = 0; i < c.size(); ++i)
    if (c[i]) return h->fail(DD_ERR_HIP, "check_finite: first non-finite tensor of this backward (stage " + std::to_string(i) + " of " + std::to_string(c.size()) + "): " +
                                          h->chk_labels[i] + ": " + std::to_string(c[i]) + " values");
  return DD_OK;
}

int check_bwd(dd_handle_t h, int precision, const char* who) {
  if (h->variant == DD_VARIANT_SWIN && precision == DD_PREC_NAIVE_FP32)
    return h->fail(DD_ERR_UNSUPPORTED, std::string(who) + ": DD_VARIANT_SWIN has no unfused path (use fp32 / bf16 / f16)");
  if (precision < DD_PREC_NAIVE_FP32 || precision > DD_PREC_LAST) return h->fail(DD_ERR_INVALID_ARG, std::string(who) + ": unknown precision");
  if (precision == DD_PREC_F16R)
    return h->fail(DD_ERR_UNSUPPORTED, std::string(who) + ": DD_PREC_F16R (refined f16) is a forward-only mode; train in fp32 / f16x3 (the forward's abs-1e-3 modes) or bf16 / f16");
  return check_split(h, precision, who);
}

}  // namespace ddapi

extern "C" {
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
  rc = check_finite_begin(h, s); if (rc) return rc;
  rc = bwd_core(h, pl, pl->x[0].as<float>(), reinterpret_cast<const long long*>(t), 0, 1, grad_cond, 0, s, kept, 0);
  if (rc) return rc;
  if (grad_x) DD_HIP(launch_nhwc_to_nchw_f32(pl->gA.p, EK_F32, grad_x, B, LATENT_C, lat_h, lat_w, 0, s));
  h->last_once_plan = pl;
  return check_finite_report(h, s);
}

}  // extern "C"

namespace ddapi {
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
}  // namespace ddapi

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
    rc = check_finite_begin(h, s); if (rc) return rc;
    rc = denoise_backward_lane(h, x_T, cond, grad_x0, grad_xT, grad_cond, B, lat_h, lat_w, cond_h, cond_w, T, precision, s, 0, 0, B, ticket, &reused, 1);
    if (rc == DD_OK && reused) h->n_traj_reuse++;
    return rc ? rc : check_finite_report(h, s);
  }
  for (int l = 1; l < S; ++l) {
    if (!h->lane_stream[l]) { rc = acquire_lane_stream(h, l, s); if (rc) return rc; }
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

}  // extern "C"
