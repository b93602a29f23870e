// dd_api_weights.cpp -- parameter intake and packing (see dd_api_internal.h for the split of the C ABI).
#include "dd_api_internal.h"

namespace ddapi {


// 0 = denoiser (model.*), 1 = latent codec (depth_transform.*), 2 = condition FPN (conv_lateral.* / conv_up.*),
// 3 = HAHI neck in front of the FPN (hahineck.*; Swin-L pyramid only)
int weight_group(const std::string& name) {
  if (name.compare(0, 6, "model.") == 0) return 0;
  if (name.compare(0, 16, "depth_transform.") == 0) return 1;
  if (name.compare(0, 9, "hahineck.") == 0) return 3;
  return 2;
}

std::vector<WeightSpec> required_weights(int variant, int pyr) {
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
}  // namespace ddapi

extern "C" {

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
}  // extern "C"

namespace ddapi {

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

}  // namespace ddapi

extern "C" {

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
}  // extern "C"
