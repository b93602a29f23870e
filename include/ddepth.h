/*
 * ddepth.h -- C ABI of the MI355X-native DDIM denoise hot path of DiffusionDepth.
 *
 * One shared library (diffusiondepth_amd/libddepth_hip.so, built by hipcc for gfx950) sits behind the
 * reference's head / pipeline seam.  Every entry point below names the reference interface it
 * replaces (paths relative to the reference tree, /root/reference).  No torch types cross this
 * boundary: plain device pointers, sizes, a hipStream_t passed as void*.
 *
 * Conventions
 *   - All tensor arguments are DEVICE pointers to contiguous fp32 NCHW data, exactly what the
 *     reference's torch tensors hold (reference runs --opt_level O0 = fp32, src/config.py:151-154).
 *     Timestep vectors are device int64 (torch.long), as in src/model/head/ddim_depth_estimate_res.py:207.
 *   - The library BORROWS caller memory for the duration of a call and writes results into
 *     caller-allocated outputs (same ownership rule as the reference's DCN extension,
 *     src/model/deformconv/src/cuda/modulated_deform_conv_cuda.cu:78,92).  It owns only its packed
 *     weights, scratch activations, schedule tables and captured hipGraphs (freed by dd_destroy).
 *   - Every function returns DD_OK (0) or a dd_status error; dd_last_error() gives the message
 *     (the reference raises C++ exceptions -> RuntimeError, modulated_deform_conv_cuda.cu:39-73; the
 *     Python shim turns a non-zero status into RuntimeError).
 *   - The library never draws random numbers: x_T / noise / timesteps are inputs (the reference draws
 *     them with torch.randn / torch.randint, ...res.py:277,203,207).
 *   - Work is enqueued on `stream` (the caller's current HIP stream, as the reference's extension
 *     uses at::cuda::getCurrentCUDAStream(), modulated_deform_conv_cuda.cu:94); calls are
 *     asynchronous.  One handle per device/process; a handle is not thread-safe.
 */
#ifndef DDEPTH_H_
#define DDEPTH_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dd_handle_s* dd_handle_t;

typedef enum dd_status {
  DD_OK = 0,
  DD_ERR_INVALID_ARG = 1,   /* bad shape / null pointer / unknown name */
  DD_ERR_HIP = 2,           /* a HIP runtime call failed (message carries hipGetErrorString) */
  DD_ERR_STATE = 3,         /* weights or schedule missing, handle misuse */
  DD_ERR_UNSUPPORTED = 4    /* valid request this build does not implement */
} dd_status;

/* Which ScheduledCNNRefine the handle implements. */
typedef enum dd_variant {
  DD_VARIANT_RES = 0,   /* src/model/head/ddim_depth_estimate_res.py:300-344  (f = cond + E[t] + NE(x)) */
  DD_VARIANT_SWIN = 1   /* src/model/head/ddim_depth_estimate_res_swin_addHAHI.py:321-382 (UpSample_add fuse) */
} dd_variant;

/* Arithmetic of the convolution contractions.  State x_t, GroupNorm statistics, schedule scalars,
 * accumulators and the decoder tail are fp32/fp64 in every mode. */
typedef enum dd_precision {
  DD_PREC_NAIVE_FP32 = 0,  /* unfused one-thread-per-output fp32 kernels: on-device cross-check      */
  DD_PREC_FP32 = 1,        /* fused implicit-GEMM on v_mfma_f32_32x32x2_f32, fp32 activations (parity gate) */
  DD_PREC_BF16 = 2,        /* fused implicit-GEMM on v_mfma_f32_32x32x16_bf16, bf16 activations (headline)  */
  DD_PREC_F16 = 3,         /* same kernels on v_mfma_f32_32x32x16_f16 (11-bit mantissa, same rate)          */
  DD_PREC_F16X3 = 4,       /* split f16: every MFMA operand an f16 pair hi + lo (~22 mantissa bits), three MFMAs per product
                              (Whi.Phi + Whi.Plo + Wlo.Phi), fp32 tensors and accumulation -- the mode that meets the
                              1e-3 ABSOLUTE depth tolerance over the whole depth range at ~1/3 of the 16-bit rate;
                              forward calls dd_denoise / dd_denoise_trace / dd_denoise_once / dd_condition; the backward calls
                              (round 6) differentiate that forward -- kept or recomputed on the split kernels: its values, ReLU
                              masks and GroupNorm statistics -- with f16 gradients (option "x3_grad_fp32" = 1: fp32 gradients) */
  DD_PREC_F16R = 5         /* refined f16: the two large convolutions on f16 operands with ONE MFMA per product, everything around
                              them made (near-)exact where it is (near-)free -- conv1 on split operands, conv3(cond) once per image
                              on split operands from the fp32 condition map, y3 and that term handed over as block-scaled int16 (f16's
                              bytes, ~15 bits), conv4's weights as an f16 pair stacked into the padding rows of its MFMA.  The 16-bit mode that holds 1e-3 depth RMSE
                              at KITTI's depth range (0..80 m) with margin; DD_VARIANT_RES, forward only                */
} dd_precision;

/* ---- lifetime ---------------------------------------------------------------------------------
 * Replaces: construction of ScheduledCNNRefine + DDIMScheduler + CNNDDIMPipiline +
 * DeepDepthTransformWithUpsampling in DDIMDepthEstimate_Res.__init__ (…res.py:36-41). */
int dd_create(dd_handle_t* out, int device, int variant);
int dd_destroy(dd_handle_t h);

/* Message of the last failing call on this handle (or, with h == NULL, of the last failing
 * dd_create on this thread).  Never NULL. */
const char* dd_last_error(dd_handle_t h);

/* Library / build identification: "ddepth <version> gfx950 ...". */
const char* dd_version(void);

/* ---- parameters -------------------------------------------------------------------------------
 * Replaces: nn.Module.load_state_dict for the keys under depth_head.model.* and
 * depth_head.depth_transform.* (key list: SURVEY.md 8a; loaded in src/main.py:422-423).
 * `name` is the state-dict key WITHOUT the "depth_head." prefix, e.g.
 * "model.noise_embedding.0.weight"; `data` is a HOST pointer to `numel` contiguous fp32 values in
 * the reference's own layout (OIHW conv weights, (in,out,kh,kw) for the ConvTranspose).
 * The condition FPN's parameters "conv_lateral.{0..3}.{0.weight,1.weight,1.bias,1.running_mean,1.running_var}" and
 * "conv_up.{0..2}.{...}" (src/model/head/ddim_depth_estimate_res.py:56-84, ..._res_swin_add.py:57-84) form a third,
 * optional group (needed by dd_condition only); lateral input widths are 64/128/256/512 for DD_VARIANT_RES and, for
 * DD_VARIANT_SWIN, 192/384/768/1536 (Swin-L) or 128/216/288/288 (MPViT-small, ..._res_mpvit_HAHI.py:32) -- recognised from numel.
 * Call dd_commit_weights after the last dd_set_weight (and again whenever weights changed, e.g.
 * after an optimizer step): it validates completeness and repacks into the kernels' layouts. */
int dd_set_weight(dd_handle_t h, const char* name, const float* data, int64_t numel);
int dd_commit_weights(dd_handle_t h, void* stream);

/* Same as dd_set_weight for parameters that already live in HBM: `data` is a DEVICE pointer (the nn.Parameter's own storage; borrowed
 * for the call, copied on `stream`).  Replaces the parameter refresh that nn.Module gets for free after optimizer.step() in the
 * reference's training loop (src/main.py:232-241): with this entry point the denoiser's weights go fp32 -> kernel layouts entirely on
 * the device at the next dd_commit_weights (pack kernels on `stream`; no host copy, no host-side packing).  Denoiser group ("model.*")
 * only: the codec and FPN groups are folded with their eval-mode BatchNorm on the host (dd_set_weight) and only run in eval mode.
 * The two routes may be mixed freely; the newest value of every parameter wins.  dd_commit_weights repacks only the groups that
 * changed since the last commit. */
int dd_set_weight_device(dd_handle_t h, const char* name, const float* data, int64_t numel, void* stream);

/* Replaces: DDIMScheduler.__init__ (src/model/diffusers/schedulers/scheduling_ddim.py:107-157).
 * `alphas_cumprod` is the HOST fp32 table of length num_train_timesteps that the scheduler built
 * (torch.cumprod(1 - linspace(beta_start, beta_end))), so table bits are the reference's own. */
int dd_set_schedule(dd_handle_t h, const float* alphas_cumprod, int num_train_timesteps);

/* ---- condition aggregation (FPN) ------------------------------------------------------------------
 * Replaces: the top-down loop of DDIMDepthEstimate_Res.forward (src/model/head/ddim_depth_estimate_res.py:108-118)
 *   x_3 = conv_lateral[3](f_3);   x_i = conv_lateral[i](f_i) + adaptive_avg_pool2d(conv_up[i](x_{i+1}), size of x_i)
 * with conv_lateral = Conv3x3(bias=False)+BN+ReLU, conv_up = ConvTranspose2d(k2,s2,bias=False)+BN+ReLU, BatchNorm in
 * eval mode (running statistics, folded into the convolutions at dd_commit_weights).
 *   feats[i]  (B, C_i, feat_h[i], feat_w[i])  device fp32 NCHW backbone features, i = 0 finest; n_levels = 4;
 *             C = {64,128,256,512} (DD_VARIANT_RES); {192,384,768,1536} or {128,216,288,288} (DD_VARIANT_SWIN, per the weights set)
 *   cond_out  (B, 256, feat_h[0], feat_w[0]) device fp32 NCHW, or NULL
 * The result also stays inside the handle in the kernels' own layout: a following dd_denoise / dd_denoise_once with
 * cond == NULL, the same B and precision and cond_h == feat_h[0], cond_w == feat_w[0] (== lat_h, lat_w for
 * DD_VARIANT_RES; for DD_VARIANT_SWIN the map is upsampled to the latent size there) uses it without any conversion.
 * precision fp32 / bf16 / f16; f16x3 / f16r: an fp32 map computed on f16-pair operands (option "cond_split"). */
int dd_condition(dd_handle_t h, const float* const* feats, const int* feat_h, const int* feat_w, int n_levels, int B,
                 float* cond_out, int precision, void* stream);

/* Replaces: `fp = self.hahineck(fp)` followed by the same top-down loop in the HAHI heads (src/model/head/
 * ddim_depth_estimate_res_swin_addHAHI.py:110-127): HAHIHeteroNeck.forward with cross_att = self_att = False, the only way any head
 * builds it (ibid. :54-56) -- per pyramid level  l = lateral_convs[i](x),  e = conv_proj / trans_proj[i-1](l)  (1x1 -> 512),
 * out = conv_fusion / trans_fusion[i-1](cat)  (3x3 over the channel concatenation; src/model/necks/hahi.py:170-173,196-197,226-272),
 * every ConvModule = bias-free conv + eval-mode BatchNorm + ReLU (folded at dd_commit_weights) -- then dd_condition's FPN on the neck's
 * outputs.  Parameters: the keys "hahineck.{lateral_convs.i, conv_proj.0, trans_proj.j, conv_fusion.0, trans_fusion.j}.{conv.weight,
 * bn.weight, bn.bias, bn.running_mean, bn.running_var}" form a fourth optional group of dd_set_weight (the neck's attention / embedding
 * parameters are never executed by the reference and are not part of it).  DD_VARIANT_SWIN with the Swin-L pyramid (192/384/768/1536) or the
 * MPViT-small one (128/216/288/288; which one shows in the sizes of the weights set).
 * Same arguments and result hand-over as dd_condition; feats are the RAW backbone maps. */
int dd_neck_condition(dd_handle_t h, const float* const* feats, const int* feat_h, const int* feat_w, int n_levels, int B,
                      float* cond_out, int precision, void* stream);

/* ---- the hot loop -----------------------------------------------------------------------------
 * Replaces: CNNDDIMPipiline.__call__ (…res.py:248-297) minus its torch.randn: `timesteps`-driven
 * loop of  eps = model(x_t, t, cond) ; x_{t-1} = scheduler.step(eps, t, x_t, eta=0,
 * use_clipped_model_output=True)  (DDIMScheduler.step, scheduling_ddim.py:231-353), for the
 * T = num_inference_steps timesteps of DDIMScheduler.set_timesteps (scheduling_ddim.py:215-229).
 *   x_T   (B,16,h,w)           initial latent noise
 *   cond  (B,256,cond_h,cond_w) condition map (cond_h,cond_w == h,w for DD_VARIANT_RES); NULL = the map the last
 *                              dd_condition call left in the handle (same B, h, w, precision)
 *   x_0   (B,16,h,w)           result ("refined_depth_t", …res.py:124)
 * The T-step loop runs as one captured hipGraph per (B,h,w,cond_h,cond_w,T,precision). */
int dd_denoise(dd_handle_t h, const float* x_T, const float* cond, float* x_0,
               int B, int lat_h, int lat_w, int cond_h, int cond_w,
               int num_inference_steps, int precision, void* stream);

/* Replaces: CNNDDIMPipiline.__call__ of the *Vis heads (src/model/head/ddim_depth_estimate_res_vis.py:284-301,
 * ..._swin_addHAHI_vis.py:289-306), which additionally returns image_list = the sample after EVERY step (decoded into the
 * 'pred_inter' output, ...res_vis.py:141-143,177).  Same arguments as dd_denoise, but
 *   states (T,B,16,h,w): states[j] = sample after step j; states[T-1] is the x_0 that dd_denoise returns.
 * Runs the T steps as individual launches (no hipGraph: every intermediate state is kept, T x 6.8 MB per KITTI image). */
int dd_denoise_trace(dd_handle_t h, const float* x_T, const float* cond, float* states, int B, int lat_h, int lat_w,
                     int cond_h, int cond_w, int num_inference_steps, int precision, void* stream);

/* Replaces: one ScheduledCNNRefine.forward(noisy_image, t, feat, None, None, None)
 * (…res.py:324-344) with per-sample timesteps t[B] (device int64), as called by ddim_loss
 * (…res.py:211).  eps (B,16,h,w) >= 0 (final GroupNorm+ReLU). */
int dd_denoise_once(dd_handle_t h, const float* x_t, const int64_t* t, const float* cond, float* eps,
                    int B, int lat_h, int lat_w, int cond_h, int cond_w, int precision, void* stream);

/* ---- training: backward of one epsilon-network evaluation -----------------------------
 * Replaces: what torch autograd does for  noise_pred = self.model(noisy_images, timesteps, *inputs)  when the reference
 * trains (ddim_loss, src/model/head/ddim_depth_estimate_res.py:201-217; loss.backward(), src/main.py:232-241): the
 * vector-Jacobian product of ScheduledCNNRefine.forward (...res.py:324-344) at (x_t, t, cond) with grad_eps (B,16,h,w).
 *   grad_x    (B,16,h,w)   dLoss/dx_t, or NULL
 *   grad_cond (B,256,h,w)  dLoss/dcond (overwritten), or NULL
 * Parameter gradients ACCUMULATE inside the handle (like Tensor.grad) under the state-dict names of dd_set_weight
 * ("model.pred.0.weight", "model.time_embedding.weight", ...) in the reference's own shapes; dd_get_grad copies one to a
 * caller DEVICE buffer, dd_zero_grad clears all.  The forward pass is recomputed internally (no stash from
 * dd_denoise_once is needed; with one -- dd_set_option "keep_trajectory" around the forward call, its "trajectory_ticket" passed through
 * "use_trajectory" in front of this call, see dd_denoise_backward -- the forward's activations are read instead).  Both variants (DD_VARIANT_SWIN: grad_cond has the condition map's own size (B,256,cond_h,cond_w),
 * the fuse convs' gradients are "model.upsample_fuse.conv{A,B}.conv.{weight,bias}"; no naive_fp32 path).  precision: naive_fp32 = unfused fp32 kernels (cross-check); fp32 / bf16 / f16 =
 * data gradients on the fused convolution kernels (transposed, flipped weights), weight gradients on the matrix cores
 * (bf16 / f16; the fp32 mode keeps the unfused weight-gradient kernel), GroupNorm backward fused into two passes per layer. */
int dd_denoise_once_backward(dd_handle_t h, const float* x_t, const int64_t* t, const float* cond, const float* grad_eps,
                             float* grad_x, float* grad_cond, int B, int lat_h, int lat_w, int cond_h, int cond_w,
                             int precision, void* stream);
/* Backward of the whole T-step loop (what autograd does for  refined_depth_t = self.pipeline(...)  in the reference's
 * training step: the loop output is NOT detached, the depth losses back-propagate through all T denoiser calls,
 * ...res.py:124-169).  Given grad_x0 = dLoss/dx_0 (B,16,h,w) it re-runs the forward loop keeping the T
 * intermediate states (16 channels each), then walks the chain x_{k+1} = c1_k x_k + c2_k eps(x_k, t_k, cond) backwards,
 * recomputing each step's activations instead of stashing them:
 *   grad_xT (B,16,h,w) or NULL;  grad_cond (B,256,h,w) = sum over the T steps, or NULL;  parameter gradients accumulate.
 * The second forward loop is skipped when the forward call already kept the states: run the training forward as
 *   dd_set_option(h, "keep_trajectory", 1); dd_denoise(...); dd_get_counter(h, "trajectory_ticket", &ticket);
 * and the backward as  dd_set_option(h, "use_trajectory", ticket); dd_denoise_backward(...)  with the same x_T / cond / shape / T /
 * precision.  The ticket is honoured only if those states are still the last thing that plan computed and no parameter was set since;
 * otherwise (and always with ticket 0 or the naive precision) the states are regenerated -- same results either way, up to the 16-bit
 * modes' hoisted condition term (the forward's own states are then the ones differentiated).  T x 16 fp32 channels per pixel are kept. */
int dd_denoise_backward(dd_handle_t h, const float* x_T, const float* cond, const float* grad_x0, float* grad_xT, float* grad_cond,
                        int B, int lat_h, int lat_w, int cond_h, int cond_w, int num_inference_steps, int precision, void* stream);
int dd_zero_grad(dd_handle_t h, void* stream);
int dd_get_grad(dd_handle_t h, const char* name, float* dst, int64_t numel, void* stream);

/* Replaces: DDIMScheduler.add_noise == q_sample (scheduling_ddim.py:355-376):
 * out = sqrt(abar[t_b]) * x0 + sqrt(1 - abar[t_b]) * noise, t (B,) device int64, tensors (B,C,h,w). */
int dd_add_noise(dd_handle_t h, const float* x0, const float* noise, const int64_t* t, float* out,
                 int B, int C, int lat_h, int lat_w, void* stream);

/* ---- latent codec -----------------------------------------------------------------------------
 * Replaces: DeepDepthTransformWithUpsampling.t (src/model/ops/depth_transform.py:29-31), eval-mode
 * BatchNorm: depth (B,1,H,W) -> latent (B,16,(H-1)/2+1,(W-1)/2+1). */
int dd_encode(dd_handle_t h, const float* depth, float* latent, int B, int H, int W, void* stream);

/* Replaces: DeepDepthTransformWithUpsampling.inv_t (depth_transform.py:33-35), eval-mode BatchNorm:
 * latent (B,16,h,w) -> depth (B,1,2h,2w) = 1/clamp(sigmoid(.),1e-6) - 1. */
int dd_decode(dd_handle_t h, const float* latent, float* depth, int B, int lat_h, int lat_w, void* stream);

/* ---- introspection (tests, bench) ----------------------------------------------------------------
 * Time of the last dd_denoise graph launch measured with hipEvents recorded on `stream` around
 * the graph (0 if timing is off).  dd_set_option("timing", 1) enables it; other options:
 * "graph" (1 = hipGraph replay, 0 = eager launches.  Default: 1 in a process whose environment had DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 when the handle was
 * created, else 0 -- the HIP 7.0 runtime's graph fast path gives wrong results in long runs of replays next to eager launches on one stream, and reads that
 * variable once, at its own initialisation: export it before the process's first HIP call; counter "graph_default" reports which default a handle took), "debug_sync" (1 = sync + check
 * after every launch), "hoist_cond" (1 = conv3(cond) and conv3(E[t]) are taken out of the DDIM loop by linearity; 0 = the
 * condition map is re-added in conv3's prologue every step; -1 [default] = hoisted in the bf16 and f16 modes of the Res variant.  Swin variant,
 * plans of the loop that keep nothing for a backward: the step-invariant part of pred.0(convB(convA(.))) -- the upsampled condition map through
 * all three convolutions once per image, the time embedding as a per-step table of border classes -- is taken out of the loop: -1 = in the 2-byte
 * modes and DD_PREC_F16X3, 1 = also in the fp32 mode, 0 = never; training plans always run the reference's order), "swin_w5" (1 [default] = those plans run
 * pred.0 o convB as ONE 5x5 convolution with a per-step border-ring correction, 0 = as two kernels (DD_PREC_F16X3: the reference's order): A/B switch), "layer_timing", "ablate" (timing experiments; honoured in -DDD_ABLATE=1
 * builds only), "naive_wgrad" (1 = backward weight gradients by the unfused kernel
 * instead of the MFMA kernel, A/B check), "x3_grad_fp32" (DD_PREC_F16X3's backward: 0 [default] = f16 gradients through the f16 mode's MFMA kernels
 * behind the split forward, 1 = fp32 gradients through the fp32 mode's kernels, ~60x slower: the parity form), "keep_trajectory" / "use_trajectory" (training: see dd_denoise_backward), "streams" (S > 1:
 * dd_denoise runs the B images as S concurrent sub-batches -- lane 0 on the caller's stream, the others on streams the handle owns, forked
 * and joined by events on the caller's stream, one plan and hipGraph per lane; the images are independent and every image's result is bit-identical to the one-lane call's as long as both take the same tile form -- since round 6 the Res denoiser's hoisted conv3 pair keeps its 8x32 tiles for every lane count (the Swin 5x5 form moves to 16x32 tiles when its 8x32 tiles exceed the resident slots, whatever the lane count), so they do; option "big_tiles" forces one form;
 * dd_denoise_backward splits the same way, with one parameter-gradient set per lane summed into the caller-visible one at the join;
 * default 1; a lane's stream is probed for concurrency with the caller's when it is created -- two idle ~100-us wavefronts, forked and joined by events --
 * and replaced when the HIP runtime put both on one hardware queue, as it does in a process that created other streams first, e.g. an eagerly
 * initialised RCCL communicator: option "lane_probe" = 0 skips the probe, counters "lane_overlap" / "lane_probe_retries" report it), "adjoint_tiled" (0 = the plain kernel for the adjoint of the Swin condition upsampling, A/B check),
 * "keep_activations_mb" (budget of the per-step activation slots kept by "keep_trajectory" forwards, default 65536: ONE figure for the
 * handle -- all lanes and shapes -- also held against the free device memory; stale sets are dropped first, and a forward that cannot
 * keep its activations keeps the states only), "thin_stream" (1 [default] = conv4 runs as the persistent streaming kernel of
 * csrc/dd_thin.hip in the 16-bit modes, 0 = as an instance of the general kernel: A/B switch), "thin_slots" (workgroups of that kernel,
 * default 512 = two per CU), "train_graphs" (0 [default] = a forward that keeps its trajectory -- a training step -- is
 * enqueued eagerly, 1 = it replays a captured hipGraph like the inference plans: round 5 measured intermittent non-finite values in 16-bit training
 * steps only with the graph; round 6 traced them to the HIP runtime's graph packet capture next to MIOpen-launched kernels -- gone with
 * DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 in the environment, profiles/r06_experiments.md section 5), "graph_fence" (diagnosis: bit 0 / 1 = a host
 * synchronisation in front of / behind every graph launch, bit 2 = the graph on the handle's own stream between two events), "check_finite" (debug: the backward counts the non-finite values of every tensor it writes and fails
 * with the name of the first one; 1 = synchronising after every stage, 2 = asynchronously, reported at the end of the call),
 * "bf16_storage" (1 = all-bf16 tensors in DD_PREC_BF16; default 0 = f16 storage),
 * "f16r_wide" (DD_PREC_F16R: 1 [default] = y3 and the hoisted conv3(cond) term travel as int16 with block scales -- one fp32 scale per pixel of y3, per
 * 32-pixel x 32-cout accumulator block of the hoisted term: f16's bytes at ~15 bits --, 0 = as f16 like in DD_PREC_F16), "f16r_c1" (DD_PREC_F16R:
 * 1 [default] = conv1's weights as an f16 pair (two MFMAs), 0 = the plain f16 kernel), "one_buffer" (1 [default] = the loop's hoisted conv3 on
 * 8x32 tiles runs in its one-patch-buffer form -- three workgroups per CU -- when a one-lane call has more tiles than resident slots; 0 = never,
 * 2 = always: A/B switch and tests),
 * "f16r_p4" (DD_PREC_F16R: 1 = conv4's operand relu(gn3(y3)) as an f16 pair as well -- two MFMAs per tap; default 0),
 * "cond_direct" (DD_PREC_F16R, DD_VARIANT_RES: 1 [default] = the once-per-image conv3(cond) reads an explicit `cond` tensor of dd_denoise* in place
 * -- NCHW fp32, eight 4-byte loads per staging item -- instead of a channel-blocked copy made first: same bits, one HBM round trip of the map less;
 * 0 = convert first: A/B switch),
 * "cond_split" (DD_PREC_F16X3 / DD_PREC_F16R: 1 [default] = dd_condition / dd_neck_condition run their convolutions on the split-f16 kernels
 * -- fp32 tensors, f16-pair operands, three MFMAs per product -- when the folded weights fit those images (counter "cond_split_ok"), 0 = on
 * the fp32-operand kernels as before round 4: A/B switch). */
int dd_set_option(dd_handle_t h, const char* key, int64_t value);
int dd_last_loop_ms(dd_handle_t h, float* ms);
/* Counters: "graph_launches", "eager_loops", "graph_capture_failures", "plans", "neck_launches", "trajectory_ticket" (ticket of the
 * last dd_denoise call that kept its states), "trajectory_reuses" (dd_denoise_backward calls that read kept states), "lane_calls" (dd_denoise calls that ran as concurrent lanes),
 * "resident_slots" (workgroup slots at two per CU), "cond_split_ok" (bit 0 / 1: the committed FPN / neck weights fit the split-f16 images). */
int dd_get_counter(dd_handle_t h, const char* key, int64_t* value);
/* With option "layer_timing" = 1 the loop runs eagerly with a hipEvent pair around every
 * convolution launch; this returns the accumulated milliseconds and launch count of conv `layer`
 * (1..4 = conv1..conv4 of the Res denoiser; 5,6,7 = convA, convB, pred.0 of the Swin variant; 9 = conv3 with the hoisted condition term; 10..13 = conv_lateral[0..3], 14 = conv_up of dd_condition; 15..18 / 24..26 = the laterals of the Swin-L / MPViT pyramids; 20..23 = data-gradient convs) since the option was set (used by bench.py for the per-kernel roofline figure). */
int dd_get_layer_ms(dd_handle_t h, int layer, double* total_ms, int64_t* launches);

/* Copies an internal intermediate of the last dd_denoise_once call to a caller DEVICE buffer as
 * fp32 NCHW: name in {"y1","y2","y3","y4"} = raw conv outputs before GroupNorm (B,C,h,w).
 * Test hook for locating a failing layer. */
int dd_debug_fetch(dd_handle_t h, const char* name, float* out, int64_t numel, void* stream);
/* Test hook: 64-bit FNV-1a digest of every packed denoiser weight buffer in HBM (the two parameter routes must agree bit for bit). */
int dd_debug_weights_digest(dd_handle_t h, uint64_t* digest);

#ifdef __cplusplus
}
#endif
#endif /* DDEPTH_H_ */
