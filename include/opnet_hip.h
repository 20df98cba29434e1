/*
 * opnet_hip.h - C ABI of libopnet_hip.so: the MI355X (gfx950) implementation of the
 * ObjectPermanence OPNet reasoner hot path.
 *
 * The reference has no FFI: its "plugin API" for this path is the PyTorch nn.Module contract
 *   ModelsFactory.get_model(name, cfg, weights) -> model;  model(boxes) -> (y_boxes, logits)
 * (reference baselines/models_factory.py:42-80, baselines/learned_models.py:18-52).  The entry
 * points below are what a binding for that contract needs; objectpermanence_amd/learned_models.py
 * is the ctypes binding that ships, INTEGRATION.md shows the stub a reference maintainer adds.
 *
 * Conventions (all entry points):
 *   - plain C types only; `stream` is a hipStream_t passed as void* (NULL = the null stream);
 *   - every pointer is a DEVICE pointer to contiguous fp32, 16-byte aligned, unless named host;
 *   - work is only ENQUEUED on `stream`; nothing synchronises the device, nothing allocates
 *     device memory (the caller owns `packed` and `workspace`, sized by the *_bytes queries);
 *   - return 0 on success, a negative OPNET_E* code otherwise; never throws.
 *     opnet_last_error() returns a thread-local message for the last failure.
 *   - shapes: 15 object slots x 6 features per frame are fixed, as in the reference
 *     (learned_models.py:21,24 hard-code `bb_in_dim * 15`); H1, H2 must be multiples of 16.
 */
#ifndef OPNET_HIP_H
#define OPNET_HIP_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OPNET_OK 0
#define OPNET_EINVAL (-1)      /* null / misaligned pointer, bad flag */
#define OPNET_ESHAPE (-2)      /* unsupported dimension */
#define OPNET_EWORKSPACE (-3)  /* packed / workspace buffer too small */
#define OPNET_EHIP (-4)        /* a HIP runtime call failed */

#define OPNET_SLOTS 15
#define OPNET_FEATS 6

int opnet_hip_abi_version(void);
const char *opnet_last_error(void);

/* ---- weights -------------------------------------------------------------------------------
 * Repack the six OPNet state_dict tensors (reference learned_models.py:29-33; names/shapes in
 * SURVEY.md 8-a1) into the MFMA-fragment order the step kernel streams:
 *   w_ih1 [4*H1, 90]  object_to_track_LSTM.weight_ih_l0     w_hh1 [4*H1, H1]  ...weight_hh_l0
 *   w_sel [15, H1]    object_to_track_prediction.weight
 *   w_ih2 [4*H2, 6]   video_LSTM.weight_ih_l0               w_hh2 [4*H2, H2]  ...weight_hh_l0
 *   w_out [4, H2]     prediction_layer.weight
 * Gate row order i,f,g,o (torch.nn.LSTM). Call again whenever the weights change. */
size_t opnet_packed_weights_bytes(int H1, int H2);
int opnet_pack_weights_f32(const float *w_ih1, const float *w_hh1, const float *w_sel,
                           const float *w_ih2, const float *w_hh2, const float *w_out,
                           float *packed, size_t packed_bytes, int H1, int H2, void *stream);

/* ---- forward (replaces OPNet.forward, learned_models.py:35-52) -------------------------------
 *   boxes  [B, T, 15, 6]   in   (the tensor Cater6TracksForObjects*Dataset yields, datasets.py:594)
 *   y      [B, T, 4]       out  (y_boxes)
 *   logits [B, 15, T]      out  (object_to_track_prediction.permute(0,2,1).contiguous(), :50)
 * Eager form: T+3 dependent step launches on `stream`. */
size_t opnet_workspace_bytes(int B, int T, int H1, int H2);
int opnet_forward_f32(const float *boxes, const float *packed, float *y, float *logits,
                      void *workspace, size_t workspace_bytes, int B, int T, int H1, int H2,
                      void *stream);

/* Graph form: the same launches as one hipGraph (host-side object; built on first use and
 * re-built only if `workspace` moves). Same results bit for bit as the eager form. */
typedef struct opnet_plan opnet_plan;
int opnet_plan_create(opnet_plan **plan, int B, int T, int H1, int H2);
int opnet_plan_forward(opnet_plan *plan, const float *boxes, const float *packed, float *y,
                       float *logits, void *workspace, size_t workspace_bytes, void *stream);
void opnet_plan_destroy(opnet_plan *plan);

/* ---- output post-processing + metric (replaces inference_main.py:219 and
 *      tracking_utils.py:137-159,251-256,278-288) ------------------------------------------------
 * y, labels [N, T, 4] fp32 normalised -> pred_px, gt_px [N, T, 4] int32 (float64 multiply by
 * [320,240,320,240], truncation toward zero), per-frame integer IoU [N, T] float64 (inclusive +1
 * pixel convention).  Any output pointer may be NULL to skip it. */
int opnet_postprocess_iou(const float *y, const float *labels, int *pred_px, int *gt_px,
                          double *iou, int N, int T, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* OPNET_HIP_H */
