/*
 * ddepth_dcn.h -- C ABI of the MI355X-native NLSPN refinement stage and of the DCNv2 operator under it
 * (same shared library as ddepth.h: diffusiondepth_amd/libddepth_hip.so).
 *
 * The reference's ONLY native code is its vendored DCNv2 CUDA extension ("DCN", src/model/deformconv/), bound with pybind11
 * (src/model/deformconv/src/vision.cpp:6-13) and used by exactly one caller, the spatial propagation of the NLSPN model
 * (src/model/nlspnmodel.py:136-141,165-171).  The entry points below are what a binding for this path attaches to:
 *   dd_dcn_forward / dd_dcn_backward      <- DCN.modulated_deform_conv_forward / _backward  (vision.cpp:10-11)
 *   dd_nlspn_offset_affinity              <- NLSPN._get_offset_affinity after its convolution (nlspnmodel.py:90-163), fused
 *   dd_nlspn_propagate                    <- the prop_time-iteration loop of NLSPN.forward    (nlspnmodel.py:186-205), fused
 * The DCNv1 (deform_conv_*) and deformable PS-RoI pooling functions of vision.cpp:8-9,12-13 are bound by no model in the reference
 * tree and are not provided.
 *
 * Conventions (those of ddepth.h): DEVICE pointers to contiguous fp32 NCHW tensors exactly as the reference's torch tensors hold
 * them (the extension dispatches float/double, src/model/deformconv/src/cuda/modulated_deform_conv_cuda.cu:91; NLSPN runs fp32);
 * inputs are borrowed, outputs are caller-allocated (the reference allocates them with at::empty / at::zeros_like, :78,:196-200);
 * work is enqueued on `stream` (at::cuda::getCurrentCUDAStream(), :92) and is asynchronous; every function returns DD_OK (0) or a
 * dd_status code and leaves the message in dd_dcn_last_error() (the reference raises through AT_ASSERTM -> RuntimeError, :39-72).
 * These functions are stateless (no handle) and thread-safe; there is no CPU path (the reference's is AT_ERROR("Not implemented
 * on the CPU"), src/model/deformconv/src/modulated_deform_conv.h:39-43).
 */
#ifndef DDEPTH_DCN_H_
#define DDEPTH_DCN_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* affinity normalisation modes of NLSPN (--affinity, src/config.py:90-95; nlspnmodel.py:48-68,101-108,150-154) */
typedef enum dd_nlspn_affinity {
  DD_AFF_AS = 0,     /* aff / (sum|aff| + 1e-4)                                   */
  DD_AFF_ASS = 1,    /* aff / max(sum|aff| + 1e-4, 1)                             */
  DD_AFF_TC = 2,     /* tanh(aff) / aff_scale_const, no normalisation             */
  DD_AFF_TGASS = 3   /* tanh(aff) / (aff_scale_const + 1e-8), then as ASS (default) */
} dd_nlspn_affinity;

/* Message of the last failing dd_dcn_* / dd_nlspn_* call on the calling thread.  Never NULL. */
const char* dd_dcn_last_error(void);

/* Replaces: modulated_deform_conv_forward (vision.cpp:10 -> modulated_deform_conv_cuda_forward,
 * modulated_deform_conv_cuda.cu:19-121): DCNv2
 *   output[b,co,ho,wo] = bias[co] + sum_{ci in group(co)} sum_{i,j} weight[co,ci,i,j] * mask[b,dg(ci),i*kw+j,ho,wo]
 *                                   * bilinear(input[b,ci], ho*stride_h - pad_h + i*dil_h + offset[b,dg(ci),2(i*kw+j),ho,wo],
 *                                                           wo*stride_w - pad_w + j*dil_w + offset[b,dg(ci),2(i*kw+j)+1,ho,wo])
 * (sampling: modulated_deform_im2col_cuda.cuh:23-54,126-194; zero outside (-1,H)x(-1,W)).
 *   input (B,C,H,W)  weight (Cout, C/group, kh, kw)  bias (Cout)  offset (B, dg*2*kh*kw, Ho, Wo)  mask (B, dg*kh*kw, Ho, Wo)
 *   output (B,Cout,Ho,Wo), Ho = (H + 2 pad_h - (dil_h (kh-1) + 1)) / stride_h + 1 (:75-76)
 * The column buffer of the reference (C*kh*kw x B*Ho*Wo floats in HBM, written by im2col and re-read by the GEMM) does not exist
 * here: sampling and contraction are one kernel.  im2col_step only chunks the reference's batch loop and does not change results;
 * it is accepted so that the reference's argument check (batch % min(batch, im2col_step) == 0, :58) fails the same way. */
int dd_dcn_forward(const float* input, const float* weight, const float* bias, const float* offset, const float* mask, float* output,
                   int B, int C, int H, int W, int Cout, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w,
                   int dilation_h, int dilation_w, int group, int deformable_group, int im2col_step, void* stream);

/* Replaces: modulated_deform_conv_backward (vision.cpp:11 -> modulated_deform_conv_cuda_backward, modulated_deform_conv_cuda.cu:124-283
 * with the kernels modulated_deform_im2col_cuda.cuh:196-328).  Any of the five outputs may be NULL (skipped); the others are
 * OVERWRITTEN with the full gradient (the reference returns fresh zero-initialised tensors):
 *   grad_input (B,C,H,W)  grad_offset like offset  grad_mask like mask  grad_weight like weight  grad_bias (Cout)
 * grad_input is scattered with fp32 atomics (as the reference's col2im does, ...cuh:248), so it is reproducible only up to fp32
 * summation order.  Like the reference's launcher (...cuh:372) the input gradient is computed with pad_w := pad_h. */
int dd_dcn_backward(const float* input, const float* weight, const float* bias, const float* offset, const float* mask,
                    const float* grad_output, float* grad_input, float* grad_offset, float* grad_mask, float* grad_weight,
                    float* grad_bias, int B, int C, int H, int W, int Cout, int kernel_h, int kernel_w, int stride_h, int stride_w,
                    int pad_h, int pad_w, int dilation_h, int dilation_w, int group, int deformable_group, int im2col_step, void* stream);

/* Replaces: NLSPN._get_offset_affinity (src/model/nlspnmodel.py:87-163) from the output of self.conv_offset_aff on, in ONE pass:
 * channel regrouping with the zero reference offset (:92-99), tanh / aff_scale_const (:101-108), the confidence of every neighbour
 * sampled at its offset -- num 1x1 modulated deformable convolutions in the reference (:114-144) --, |.|-sum normalisation
 * (:146-154) and the reference-pixel affinity 1 - sum (:156-161).
 *   offset_aff      (B, 3*num, H, W), num = k_f*k_f - 1          output of conv_offset_aff
 *   confidence      (B, 1, H, W) or NULL (conf_prop == 0)
 *   aff_scale_const device pointer to the 1-element parameter (:59-68)
 *   w_conf, b_conf  device pointers to NLSPN.w_conf (1 element) and NLSPN.b (1 element), the weight / bias of the 1x1 sampling (:79-80,75)
 *   offset          (B, 2*(num+1), H, W)   out         aff (B, num+1, H, W)   out
 *   legacy != 0 reproduces --legacy (:126-134), including its in-place effect on the returned offsets.  k_f in {3, 5, 7}. */
int dd_nlspn_offset_affinity(const float* offset_aff, const float* confidence, const float* aff_scale_const, const float* w_conf,
                             const float* b_conf, float* offset, float* aff, int B, int H, int W, int k_f, int affinity, int conf_prop,
                             int legacy, void* stream);

/* The same stage INCLUDING the convolution in front of it: offset_aff = self.conv_offset_aff(guidance) (nlspnmodel.py:90,
 * nn.Conv2d(ch_g, 3*num, k_g, padding (k_g-1)/2, bias), :50-53) is evaluated inside the kernel and the 3*num-plane intermediate never
 * reaches HBM.  Built for the geometry NLSPNModel uses (ch_g = 8, k_g = 3, k_f = 3; nlspnmodel.py:215,287-288); anything else is
 * DD_ERR_UNSUPPORTED (run the convolution separately and call dd_nlspn_offset_affinity).
 *   guidance (B, ch_g, H, W)   conv_weight (3*num, ch_g, k_g, k_g)   conv_bias (3*num)   -- device pointers; the rest as above. */
int dd_nlspn_guided_offset_affinity(const float* guidance, const float* conv_weight, const float* conv_bias, const float* confidence,
                                    const float* aff_scale_const, const float* w_conf, const float* b_conf, float* offset, float* aff,
                                    int B, int ch_g, int H, int W, int k_g, int k_f, int affinity, int conf_prop, int legacy, void* stream);

/* Bytes of scratch dd_nlspn_propagate needs (0 unless preserve_input). */
int dd_nlspn_workspace_bytes(int B, int H, int W, int preserve_input, int64_t* bytes);

/* Replaces: the propagation loop of NLSPN.forward (nlspnmodel.py:186-205): prop_time times
 *   [feat = (1 - mask_fix) * feat + mask_fix * feat_fix   if preserve_input (:199-201), mask_fix = feat_fix > 0 (:189-191)]
 *   feat = ModulatedDeformConvFunction(feat, offset, aff, w, b, stride 1, padding (k_f-1)/2, dilation 1, groups 1, 1, 64)  (:165-171)
 * for the one-channel depth map (ch_f == 1, asserted by the reference, :30).
 *   feat_init (B,1,H,W)   offset (B,2*k_f*k_f,H,W)   aff (B,k_f*k_f,H,W)   feat_fix (B,1,H,W) or NULL
 *   w (k_f*k_f) and b (1): device pointers to the "dummy parameters for gathering" NLSPN.w / NLSPN.b (:74-75)
 *   feat_list (prop_time, B, 1, H, W) out: the result of EVERY iteration (the reference returns them as list_feat -> output
 *             'pred_inter', :203,:207); the final feat_result is feat_list[prop_time-1]
 *   workspace: dd_nlspn_workspace_bytes() bytes of device scratch (may be NULL when that is 0)
 * One launch per iteration (an iteration reads its whole predecessor): 27 planes in, 1 out per pixel, HBM-bound. */
int dd_nlspn_propagate(const float* feat_init, const float* offset, const float* aff, const float* feat_fix, const float* w,
                       const float* b, float* feat_list, void* workspace, int B, int H, int W, int k_f, int prop_time,
                       int preserve_input, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DDEPTH_DCN_H_ */
