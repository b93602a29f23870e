/*
 * ddepth_msda.h -- C ABI of the multi-scale deformable attention operator of the HAHI neck (same shared library as
 * ddepth.h: diffusiondepth_amd/libddepth_hip.so).
 *
 * The reference's HAHI neck builds two mmcv.ops.MultiScaleDeformableAttention modules (src/model/necks/hahi.py:10,108-118) and calls them
 * for the hierarchical self attention over the transformer levels (hahi.py:211-223) and for the cross attention of the convolutional
 * level onto them (hahi.py:235-247).  The module's core is ONE native operator of the un-vendored dependency mmcv-full
 * (requirements.txt:84 pins 1.3.13, README.md:78 names 1.6.2; absent from /root/reference and from this image):
 *     MultiScaleDeformableAttnFunction.apply(value, spatial_shapes, level_start_index, sampling_locations, attention_weights, im2col_step)
 * = ext_module.ms_deform_attn_forward / ms_deform_attn_backward -- the operator of Deformable DETR (Zhu et al., ICLR 2021, eq. 3).  The two
 * entry points below are what a binding of that pair attaches to; the arithmetic is restated in oracle/msda_oracle.py (PARITY UNPINNED: no
 * reference-side run exists -- every DiffusionDepth head builds the neck with cross_att = self_att = False, and with attention on the neck
 * as constructed there, 3 transformer levels against num_levels = 4, cannot broadcast; see the oracle's header).
 *
 * Conventions (those of ddepth_dcn.h): DEVICE pointers to contiguous fp32 tensors in mmcv's layouts; inputs borrowed, outputs
 * caller-allocated; work enqueued on `stream`, asynchronous; returns DD_OK (0) or a dd_status code with the message in
 * dd_msda_last_error(); stateless, thread-safe, no CPU path.
 */
#ifndef DDEPTH_MSDA_H_
#define DDEPTH_MSDA_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Message of the last failing dd_msda_* call on the calling thread.  Never NULL. */
const char* dd_msda_last_error(void);

/* Replaces: mmcv ext_module.ms_deform_attn_forward (called at mmcv/ops/multi_scale_deform_attn.py, MultiScaleDeformableAttnFunction.forward;
 * reference call sites hahi.py:211-223,235-247):
 *   out[b, q, m, c] = sum_{l < L} sum_{p < P}  w[b, q, m, l, p] * bilinear(value_l[b, :, m, c],  y = loc_y * H_l - 0.5,  x = loc_x * W_l - 0.5)
 * with value_l the H_l x W_l map of level l (rows level_start_index[l] .. + H_l * W_l of `value`), zero outside the map (every corner of
 * the bilinear cell is tested on its own; a sample whose centre is not inside (-1, H_l) x (-1, W_l) contributes nothing).
 *   value               (B, num_keys, M, D)      num_keys = sum_l H_l * W_l, M heads, D channels per head
 *   spatial_shapes      (L, 2) int64, rows (H_l, W_l)        level_start_index (L) int64
 *   sampling_locations  (B, Q, M, L, P, 2), (x, y) normalised to [0, 1] over the level's map
 *   attention_weights   (B, Q, M, L, P)
 *   out                 (B, Q, M * D)
 * im2col_step only chunks mmcv's batch loop (B % min(B, im2col_step) must be 0, checked as there); results do not depend on it. */
int dd_msda_forward(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index, const float* sampling_locations,
                    const float* attention_weights, float* out, int B, int num_keys, int M, int D, int L, int Q, int P, int im2col_step,
                    void* stream);

/* Replaces: mmcv ext_module.ms_deform_attn_backward.  grad_out (B, Q, M * D).  Any of the three outputs may be NULL (skipped); the others
 * are OVERWRITTEN with the full gradient: grad_value like value (scattered with fp32 atomics, as mmcv's col2im does: reproducible up to
 * summation order), grad_sampling_loc like sampling_locations, grad_attn_weight like attention_weights. */
int dd_msda_backward(const float* value, const int64_t* spatial_shapes, const int64_t* level_start_index, const float* sampling_locations,
                     const float* attention_weights, const float* grad_out, float* grad_value, float* grad_sampling_loc,
                     float* grad_attn_weight, int B, int num_keys, int M, int D, int L, int Q, int P, int im2col_step, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DDEPTH_MSDA_H_ */
