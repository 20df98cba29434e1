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
#include <stdint.h>

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

/* bumped whenever entry points are added or a signature changes; the Python mirror refuses a library of another version
 * (objectpermanence_amd/_lib.py) instead of failing later on a missing symbol.  4 = round 4; 8 = the Winograd entry points (opdet_wino_*);
 * 9 = opseq_slot_embed_bwd_workspace_bytes / opseq_slot_embed_relu_bwd_ws_f32. */
#define OPNET_HIP_ABI_VERSION 9
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

/* Per-XCD persistent form (H1 = 256, H2 = 512 only - configs/opnet_model_config.json): ONE launch in which each of the
 * chip's 8 XCDs runs the whole T-step forward of its own share of the clips (groups of 16) with every weight resident in
 * registers; the only exchange is h1 / h2 among the 32 CUs of one XCD.  Same inputs / outputs as opnet_forward_f32 and
 * the same `packed` image; results agree with the step-launch form to fp32 rounding (different K-split summation
 * order).  B <= opnet_xcd_max_batch() clips per call; throughput needs >= 2 groups per XCD (B >= 256), a small batch is
 * faster through opnet_plan_forward.  The workspace keeps the h1 / h2 history (~1 MB per clip at T = 300); its first 12
 * bytes are status words {abort code, block, phase}: a launch that could not complete (a workgroup not resident for
 * ~1.5 s) sets the code and fills y with NaN instead of hanging.  Launches of this form are chained per device through
 * an event (two persistent grids must not be co-resident), whatever streams the callers use. */
int opnet_xcd_max_batch(void);
/* 1 when the persistent form can run on the current device for these sizes (H1 = 256, H2 = 512, all 256 CUs visible) */
int opnet_xcd_supported(int H1, int H2);
size_t opnet_xcd_workspace_bytes(int B, int T, int H1, int H2);
int opnet_xcd_forward_f32(const float *boxes, const float *packed, float *y, float *logits,
                          void *workspace, size_t workspace_bytes, int B, int T, int H1, int H2,
                          void *stream);
/* the same launch over `nreq` (<= 64) request tensors boxes[r] = [counts[r]][T][90] (device pointers in a HOST array; the table
 * travels in the kernel arguments): the launch's clips are the requests' clips in order, y / logits are [sum counts][..] - a
 * server's pending requests (serving.ReasonerServer; the reference feeds its model one DataLoader minibatch at a time,
 * inference_main.py:191-217) without a concatenation copy */
int opnet_xcd_forward_multi_f32(const float *const *boxes, const int *counts, int nreq, const float *packed, float *y,
                                float *logits, void *workspace, size_t workspace_bytes, int T, int H1, int H2, void *stream);
/* measurement: with profiling enabled every launch of the persistent kernel (not the input pack / output head around
 * it) is bracketed by HIP events on the caller's stream; opnet_xcd_profile_read waits for them (host sync) and returns
 * the summed kernel time and the number of launches since the last read. */
int opnet_xcd_profile(int enable);
int opnet_xcd_profile_read(double *kernel_ms_total, int *launches);
/* the same for the other profiled kernels while opnet_xcd_profile(1) is on: tag 0 = opnet_xcd_forward, 1 = seqx_forward (the
 * persistent stacked LSTM), 2 = the attention kernel(s) of an encoder layer's attention call, 3 = seqt_forward, 4 = the fused
 * feed-forward kernel of an encoder layer, 5 / 6 = the flash attention launches of an encoder layer's TRAINING forward / backward
 * (csrc/attn_train_kernels.hip), 7 = seqx_backward (the stacked LSTM's reverse recurrence as one launch) */
int opnet_kernel_profile_read(int tag, double *kernel_ms_total, int *launches);
/* tools: device buffer of >= (T+1) * ceil(B/128) * 8 uint64 receiving s_memtime stamps of block 0 (NULL = off) */
void opnet_xcd_set_trace(void *device_buffer);
/* ---- one small request as ONE persistent launch: groups of FOUR clips, one per XCD (csrc/opnet_xcd4_kernels.hip) -------------
 * The same function as opnet_forward_f32 for B <= opnet_xcd4_max_batch() clips (meant for B <= 64: the reference's inference
 * batch_size is 16, configs/inference_config.json:2, its training batch 32; row blocks of 32 clips run one after the other) at the reference hidden sizes on a whole MI355X
 * (opnet_xcd_supported): every weight resident in registers for all T steps, h exchanged through the XCD's L2 with sentinel-armed
 * rings (no flags).  A 32-clip forward takes 0.75 ms against 1.2 ms through the launch chain.  `packed` is its own image
 * (opnet_xcd4_pack_weights_f32); the workspace holds the packed input, the h2 history (for the output head) and the rings.
 * Same error / stream / never-allocate conventions as above; an aborted launch (bounded polls) leaves NaN in y. */
int opnet_xcd4_max_batch(void);
size_t opnet_xcd4_packed_weights_bytes(int H1, int H2);
size_t opnet_xcd4_workspace_bytes(int B, int T, int H1, int H2);
int opnet_xcd4_pack_weights_f32(const float *w_ih1, const float *w_hh1, const float *w_sel, const float *w_ih2,
                                const float *w_hh2, const float *w_out, float *packed, size_t packed_bytes,
                                int H1, int H2, void *stream);
int opnet_xcd4_forward_f32(const float *boxes, const float *packed, float *y, float *logits, void *workspace,
                           size_t workspace_bytes, int B, int T, int H1, int H2, void *stream);
/* tools: the same for the 4-clip persistent training kernels (opnet_train_forward_f32 / opnet_train_backward_f32 on
 * batches of up to 32 clips): >= (T+3) * ceil(B/32) * 8 uint64 */
void opnet_xcd4_set_trace(void *device_buffer);
/* tools / tests: status words of this process's most recent 4-clip persistent launch (synchronises the device): [0] abort
 * code (0 = ok), [1] first failing block, [2] phase, [3] groups that ran the write-through protocol (not XCD-local) */
int opnet_xcd4_last_status(unsigned *out4);
/* Where a persistent launch keeps its 4 status words inside the caller's workspace (byte offset; (size_t)-1 = this shape never
 * runs a persistent kernel), so that a host caller can mirror them behind the launch (one 16-byte async copy to pinned memory)
 * and re-run an aborted batch on the launch chain instead of consuming NaN.  opnet_xcd_forward_f32: offset 0.  The training
 * words are sticky from opnet_train_forward_f32 to opnet_train_backward_f32 of the same step (either may raise them); after
 * an abort the weight gradients are NaN and a guarded Adam step (below) leaves the parameters untouched. */
size_t opnet_xcd4_status_offset(int B, int T, int H1, int H2);
size_t opnet_train_status_offset(int B, int T, int H1, int H2);
/* run-time switch of the 4-clip persistent kernels (training step and small inference request): 0 = launch chain only.
 * Replaces nothing in the reference; it is how a caller falls back after an aborted persistent launch. */
void opnet_xcd4_enable(int on);
int opnet_xcd4_enabled(void);

/* ---- training (replaces torch autograd through OPNet.forward, nn.L1Loss and torch.optim.Adam as used
 *      at training_main.py:150-152,183-217) ---------------------------------------------------------
 * opnet_train_forward_f32 is opnet_forward_f32 that additionally keeps every step's h, c, gates,
 * slot probabilities and frames_boxes in `workspace` (5.7 MB/clip at T=300, plus 64 MB whatever the batch: the partial
 * tiles of the weight-gradient waves).
 * opnet_train_backward_f32 consumes that history and dy = dLoss/dy_boxes [B,T,4] and writes the six
 * weight gradients in the state_dict layouts (`boxes` never requires grad; the logits output is not
 * differentiated - no reference loss uses it, training_main.py:186-210).  `packed` must come from
 * opnet_train_pack_weights_f32 (inference tiles + transposed tiles for the backward recurrence).
 * One workspace holds ONE forward's history: call backward before the next train forward.
 * On a whole MI355X at the reference hidden sizes batches of up to 32 clips run both recurrences as one persistent launch each
 * (4-clip groups, one per XCD) - same arguments, same histories.  There opnet_train_pack_weights_f32 packs only what
 * those launches read and leaves the launch chain's layouts to the first forward / backward that needs them: the six weight
 * pointers must stay valid and unchanged from the pack call until that step's backward has been enqueued (they are anyway:
 * the optimiser runs after it). */
size_t opnet_train_packed_weights_bytes(int H1, int H2);
int opnet_train_pack_weights_f32(const float *w_ih1, const float *w_hh1, const float *w_sel,
                                 const float *w_ih2, const float *w_hh2, const float *w_out,
                                 float *packed, size_t packed_bytes, int H1, int H2, void *stream);
size_t opnet_train_workspace_bytes(int B, int T, int H1, int H2);
int opnet_train_forward_f32(const float *boxes, const float *packed, float *y, float *logits,
                            void *workspace, size_t workspace_bytes, int B, int T, int H1, int H2,
                            void *stream);
int opnet_train_backward_f32(const float *dy, const float *packed, void *workspace, size_t workspace_bytes,
                             float *g_ih1, float *g_hh1, float *g_sel, float *g_ih2, float *g_hh2,
                             float *g_out, int B, int T, int H1, int H2, void *stream);
/* loss = mean(|y - labels|) over n elements (nn.L1Loss(reduction="none") + torch.mean); dy (may be NULL)
 * = sign(y - labels) / n.  scratch: >= 4096 bytes of device memory. Deterministic reduction. */
int opnet_l1_loss_f32(const float *y, const float *labels, float *loss, float *dy, long n, void *scratch,
                      size_t scratch_bytes, void *stream);
/* torch.nn.SmoothL1Loss(beta) with mean reduction - the loss BASELINE.json's config text names; the
 * reference itself trains with L1 (training_main.py:152). */
int opnet_smooth_l1_loss_f32(const float *y, const float *labels, float *loss, float *dy, long n, float beta,
                             void *scratch, size_t scratch_bytes, void *stream);
/* One torch.optim.Adam step on one tensor (no weight decay / amsgrad); `step` counts from 1;
 * the gradient is multiplied by grad_scale first (1/world for data-parallel sums). */
int opnet_adam_step_f32(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, long n, float lr,
                        float beta1, float beta2, float eps, int step, float grad_scale, void *stream);
/* the same update for `count` (<= 16) tensors that share the hyper-parameters and the step number, as ONE launch (a model's
 * parameters: torch.optim.Adam's per-parameter loop, training_main.py:217); arrays of `count` device pointers / element counts */
int opnet_adam_multi_step_f32(int count, float *const *params, const float *const *grads, float *const *exp_avgs,
                              float *const *exp_avg_sqs, const long *numels, float lr, float beta1, float beta2,
                              float eps, int step, float grad_scale, void *stream);
/* the same with device-side guards (each may be NULL): the whole update is skipped - parameters and moments untouched, no host
 * round trip - when *abort_u32 != 0 (status word of the persistent launches that produced the gradients), when *loss_f32 is
 * not finite, or when guard_f32[0] or guard_f32[1] != 0 (data parallel: the sums over ranks of their abort / non-finite-loss flags,
 * carried by the gradient all-reduce; opnet_dp_guard_f32).  torch.optim.Adam (training_main.py:217) has no such guard: it would write NaN into every weight. */
int opnet_adam_multi_step_guarded_f32(int count, float *const *params, const float *const *grads, float *const *exp_avgs,
                                      float *const *exp_avg_sqs, const long *numels, float lr, float beta1, float beta2,
                                      float eps, int step, float grad_scale, const unsigned *abort_u32, const float *loss_f32,
                                      const float *guard_f32, void *stream);
/* data parallel: this rank's guard words, written on `stream` into the 4 floats the caller keeps behind its flat gradient
 * bucket and all-reduces (sum) WITH the gradients: [0] = 1 if *abort_u32 != 0, [1] = 1 if *loss_f32 is not finite, [2] =
 * *loss_f32 * loss_weight (n_local / n_global: the sum is the mean loss of the whole minibatch, training_main.py:212, without
 * a collective of its own), [3] = 0; either pointer may be NULL (= 0).  After the sum `guard_f32` of the guarded Adam above points at them: nonzero [0] or [1]
 * skips the update on EVERY rank alike (a rank-local skip would let weights and moments drift apart between the ranks). */
int opnet_dp_guard_f32(float *guard4_f32, const unsigned *abort_u32, const float *loss_f32, float loss_weight, void *stream);

/* ---- input encoder (host code; replaces baselines/datasets.py:130-196 / :265-336 _normalize_and_pad_predictions and
 *      :199-257 / :338-416 _get_closest_object_to_track_vector) -------------------------------------------------------
 * Per-frame detections of n_clips clips -> boxes [n_clips][T][15][n_tracks] fp32 (slot order: snitch 140 first, then ascending
 * class id; first detection of an id per frame, last for a repeated snitch; [x1/320, y1/240, x2/320, y2/240, 1(, is_cone)];
 * a missing cone keeps its cone bit below the frame's largest rank) and the heuristic object-to-track vector [n_clips][T]
 * (int64; may be NULL).  HOST pointers: counts [frames] detections per frame, ids [N], bb [N][4] pixels, clip c = frames
 * clip_first_frame[c] .. + T and detections clip_first_det[c] .. clip_first_det[c + 1]; is_cone [n_classes] = the class table
 * (object_indices.py:200-202).  n_tracks 5 or 6.  Returns 0 / negative error code; touches no GPU state. */
int opnet_encode_clips_f32(const int32_t *counts, const int32_t *ids, const int32_t *bb, const int64_t *clip_first_frame,
                           const int64_t *clip_first_det, int n_clips, int T, int n_tracks, const uint8_t *is_cone, int n_classes,
                           float *boxes_out, int64_t *index_out);
/* The reference's clip FILES read by native host code (replaces pickle.load at baselines/datasets.py:60-64, 583-587 and the
 * json.load + label arithmetic of :33-45) and encoded as above, n_clips at a time: pkl_paths[c] = <video>.pkl as
 * preprocess_perception_main.py:87-96 writes it - a pickled dict whose "bb" / "labels" entries are lists of T numeric ndarrays
 * [n_t, 4] / [n_t] (protocols 2-5; other keys are parsed and ignored) - read by a RESTRICTED unpickler: a stack machine over
 * the opcodes such a file uses with a whitelist of numpy's ndarray / dtype reconstruction callables; nothing is imported or
 * executed, anything else is refused.  json_paths[c] (array or entries may be NULL) = <video>_bb.json: labels_out [c][T][4] =
 * the snitch's [x, y, x + w, y + h] / [320, 240, 320, 240] (float64 division, fp32 cast).  HOST pointers; boxes_out
 * [n_clips][T][15][n_tracks], index_out [n_clips][T] (may be NULL), labels_out (may be NULL).  Returns 0, or -5 (a file cannot be
 * read), -6 (refused / malformed), -7 (not T frames), -1 / -3 (arguments) with a message naming the file in err[err_len]. */
int opnet_load_clips_f32(const char *const *pkl_paths, const char *const *json_paths, int n_clips, int T, int n_tracks,
                         const uint8_t *is_cone, int n_classes, float *boxes_out, int64_t *index_out, float *labels_out, char *err,
                         int err_len);

/* ---- sibling reasoners (reference learned_models.py:55-197) ----------------------------------------
 * OPNetLstmMlp (:55-89): OPNet whose video LSTM is relu(Linear 6->H2) (hidden_layer.weight [H2,6]);
 * same packed/workspace sizes as OPNet (opnet_packed_weights_bytes / opnet_workspace_bytes). */
int opnet_mlp_pack_weights_f32(const float *w_ih1, const float *w_hh1, const float *w_sel,
                               const float *w_hidden, const float *w_out, float *packed,
                               size_t packed_bytes, int H1, int H2, void *stream);
int opnet_mlp_forward_f32(const float *boxes, const float *packed, float *y, float *logits,
                          void *workspace, size_t workspace_bytes, int B, int T, int H1, int H2, void *stream);

/* OPNetLstmMlp training (reference learned_models.py:55-89 under training.py:17-150): same history workspace,
 * sizes and conventions as the opnet_train_* entry points (opnet_train_packed_weights_bytes /
 * opnet_train_workspace_bytes give the sizes).  hidden_layer.weight [H2,6] rides in the gate-0 rows of the
 * OPNet "W_ih2" block, so `scratch` (>= 4*H2*6 floats, device) is needed by pack and receives the padded
 * gradient in backward: rows [0,H2) of g_hidden_scratch are d hidden_layer.weight, the rest is zero. */
int opnet_mlp_train_pack_weights_f32(const float *w_ih1, const float *w_hh1, const float *w_sel,
                                     const float *w_hidden, const float *w_out, float *packed, size_t packed_bytes,
                                     float *scratch4h2x6, int H1, int H2, void *stream);
int opnet_mlp_train_forward_f32(const float *boxes, const float *packed, float *y, float *logits, void *workspace,
                                size_t workspace_bytes, int B, int T, int H1, int H2, void *stream);
int opnet_mlp_train_backward_f32(const float *dy, const float *packed, void *workspace, size_t workspace_bytes,
                                 float *g_ih1, float *g_hh1, float *g_sel, float *g_hidden_scratch, float *g_out,
                                 int B, int T, int H1, int H2, void *stream);
/* L (1..3) stacked bias-free LSTM layers (input width KX, hidden H each) + Linear H->4:
 *   x [B,T,KX] -> y [B,T,4].  BaselineLstm (:92-118): L=1, KX=75.  NonLinearLstm (:121-151): L=2,
 *   KX=15*F after opseq_slot_embed_relu_f32.  TransformerLstm's LSTM half (:170-172,192-195): L=2, KX=E.
 * w_ih / w_hh are HOST arrays of L device pointers (video_LSTM.weight_ih_l{k} [4H,K_k], weight_hh_l{k}
 * [4H,H]); w_head = predictions_layer.weight [4,H]. */
size_t opseq_lstm_stack_packed_bytes(int L, int KX, int H);
size_t opseq_lstm_stack_workspace_bytes(int B, int T, int L, int KX, int H);
int opseq_lstm_stack_pack_weights_f32(const float *const *w_ih, const float *const *w_hh, const float *w_head,
                                      float *packed, size_t packed_bytes, int L, int KX, int H, void *stream);
int opseq_lstm_stack_forward_f32(const float *x, const float *packed, float *y, void *workspace,
                                 size_t workspace_bytes, int B, int T, int L, int KX, int H, void *stream);
/* same, with the T+L step launches replayed from a hipGraph cached per (workspace, packed, shape) inside
 * the library (host-side objects; opseq_graph_cache_clear() releases them). Bit-identical results. */
int opseq_lstm_stack_forward_graph_f32(const float *x, const float *packed, float *y, void *workspace,
                                       size_t workspace_bytes, int B, int T, int L, int KX, int H, void *stream);
/* ---- the same stack as ONE persistent launch (csrc/seq_xcd_kernels.hip) --------------------------------------------------
 * Replaces the T + 2L - 1 step launches above for the reference's three stacked reasoners (learned_models.py:99-101,
 * 135-137, 170-171: H = 512; L = 1 with KX = 75, L = 2 with KX = 256 or the hoisted KX = 3840) on a whole MI355X
 * (opseq_xcd_supported): every weight of a layer resident in the registers of one XCD for all T steps, groups of four clips,
 * layer 1 on the neighbouring XCD one or more steps behind layer 0.  B <= opseq_xcd_max_batch(L) clips per launch.  `packed`
 * is its own image (opseq_xcd_pack_weights_f32: LSTM weights only); w_head is predictions_layer.weight [4][H] where the caller
 * holds it.  Same error / stream / never-allocate conventions; an aborted launch (bounded polls) leaves NaN in y and raises
 * the status words at opseq_xcd_status_offset() of the workspace. */
int opseq_xcd_supported(int L, int KX, int H);
void opseq_xcd_enable(int on);
int opseq_xcd_max_batch(int L);
size_t opseq_xcd_packed_bytes(int L, int KX, int H);
size_t opseq_xcd_workspace_bytes(int B, int T, int L, int KX, int H);
size_t opseq_xcd_status_offset(int B, int T, int L, int KX, int H);
int opseq_xcd_pack_weights_f32(const float *const *w_ih, const float *const *w_hh, float *packed, size_t packed_bytes,
                               int L, int KX, int H, void *stream);
int opseq_xcd_forward_f32(const float *x, const float *packed, const float *w_head, float *y, void *workspace,
                          size_t workspace_bytes, int B, int T, int L, int KX, int H, void *stream);
/* ---- the same stack as ONE persistent launch, THROUGHPUT form (csrc/seq_xcdt_kernels.hip) -----------------------------------
 * For batches (the inference driver over a dataset, the per-epoch evaluation, a server's merged passes): groups of SIXTEEN clips
 * on v_mfma_f32_16x16x4_f32, a product wave and a finish wave per SIMD (the scheme of opnet_xcd_forward_f32).  H = 512;
 * L = 1 with KX <= 80 (learned_models.py:99-101: one copy of the layer per XCD) or L = 2 with KX % 16 == 0 (:135-137, :170-171:
 * the layer-0 input product as one GEMM before the launch, the two layers on a pair of XCDs).  At most
 * opseq_xcdt_max_batch(T, ...) clips per launch (callers loop).  Results agree with opseq_xcd_forward_f32 /
 * opseq_lstm_stack_forward_f32 to rounding (another summation order), and are reproduced bit for bit from run to run.
 * Same conventions as above (own packed image, w_head where the caller holds it, status words, NaN in y after an abort). */
int opseq_xcdt_supported(int L, int KX, int H);
void opseq_xcdt_enable(int on);
int opseq_xcdt_max_batch(int T, int L, int KX, int H);
size_t opseq_xcdt_packed_bytes(int L, int KX, int H);
size_t opseq_xcdt_workspace_bytes(int B, int T, int L, int KX, int H);
size_t opseq_xcdt_status_offset(int B, int T, int L, int KX, int H);
int opseq_xcdt_pack_weights_f32(const float *const *w_ih, const float *const *w_hh, float *packed, size_t packed_bytes,
                                int L, int KX, int H, void *stream);
int opseq_xcdt_forward_f32(const float *x, const float *packed, const float *w_head, float *y, void *workspace,
                           size_t workspace_bytes, int B, int T, int L, int KX, int H, void *stream);
/* tools: device buffer of >= 2 * (T + 1) * 8 * 8 uint64 receiving s_memtime stamps of blocks 0 and 1 (NULL = off) */
void opseq_xcdt_set_trace(void *device_buffer);
/* opseq_lstm_stack_train_forward_f32 runs the same persistent launch on the shapes above (B <= opseq_xcd_max_batch(L)) and
 * writes the h / c / gate histories its backward reads; its status words sit at this offset of the TRAINING workspace
 * ((size_t)-1: this shape trains on the launch chain) */
size_t opseq_lstm_stack_train_status_offset(int B, int T, int L, int KX, int H);
/* tools: device buffer of >= 8 * 4 * (T * groups per XCD) * 8 uint64 receiving s_memtime stamps of CU 0 of every XCD (NULL = off) */
void opseq_xcd_set_trace(void *device_buffer);

void opseq_graph_cache_clear(void);
/* training of the stacked LSTM: forward keeping the history in `workspace`, then BPTT + weight gradients.
 * g_ih / g_hh: HOST arrays of L device pointers (state_dict layouts); dx0 (may be NULL): gradient of x [B,T,KX]. */
size_t opseq_lstm_stack_train_packed_bytes(int L, int KX, int H);
size_t opseq_lstm_stack_train_workspace_bytes(int B, int T, int L, int KX, int H);
int opseq_lstm_stack_train_pack_weights_f32(const float *const *w_ih, const float *const *w_hh, const float *w_head,
                                            float *packed, size_t packed_bytes, int L, int KX, int H, void *stream);
int opseq_lstm_stack_train_forward_f32(const float *x, const float *packed, float *y, void *workspace,
                                       size_t workspace_bytes, int B, int T, int L, int KX, int H, void *stream);
int opseq_lstm_stack_train_backward_f32(const float *dy, const float *packed, void *workspace, size_t workspace_bytes,
                                        float *const *g_ih, float *const *g_hh, float *g_head, float *dx0, int B,
                                        int T, int L, int KX, int H, void *stream);
/* gradient of boxes_linear.weight [F,5] given the saved forward output and its gradient */
int opseq_slot_embed_relu_bwd_f32(const float *x, const float *out, const float *dout, float *dW, long ntok,
                                  int nslots_out, int F, void *stream);
/* the same with a caller workspace (opseq_slot_embed_bwd_workspace_bytes): rows of out / dout read coalesced, per-block partial sums
 * added in block order (deterministic) - 0.59 -> 0.08 ms at 9 600 tokens x 15 slots x 256 features */
size_t opseq_slot_embed_bwd_workspace_bytes(long ntok, int nslots_out, int F);
int opseq_slot_embed_relu_bwd_ws_f32(const float *x, const float *out, const float *dout, float *dW, long ntok,
                                     int nslots_out, int F, void *workspace, size_t workspace_bytes, void *stream);
/* relu(boxes_linear(x)) (:138,:178): x [ntok,15,5], W [F,5] -> out [ntok, nslots_out, F];
 * nslots_out = 15 (all slots) or 1 (slot 0 only - the live path of TransformerLstm). */
int opseq_slot_embed_relu_f32(const float *x, const float *W, float *out, long ntok, int nslots_out, int F,
                              void *stream);
/* attention core of nn.MultiheadAttention (learned_models.py:166-168 via nn.TransformerEncoderLayer) over ONE sequence:
 * qkv [S][3E] = (q | k | v) after the input projection -> out [S][E]; head size E/nhead a multiple of 16, <= 128. */
size_t opseq_attention_workspace_bytes(long S, int E, int nhead);
/* workspace (nullable): scratch for the key-split partials; without it the kernel runs unsplit */
int opseq_attention_f32(const float *qkv, float *out, long S, int E, int nhead, void *workspace,
                        size_t workspace_bytes, void *stream);
/* One post-LN nn.TransformerEncoderLayer (eval mode, ReLU FFN, eps 1e-5; :166-168,184) applied IN PLACE
 * to ONE sequence z [S,E] (S = B*T: the reference's sequence-first call attends across all frames
 * of the minibatch). Parameters in state_dict layouts: in_proj [3E,E]+[3E], out_proj [E,E]+[E],
 * linear1 [ffn,E]+[ffn], linear2 [E,ffn]+[E], norm1/norm2 weight+bias [E]. E and E/nhead multiples of 16. */
size_t opseq_encoder_workspace_bytes(long S, int E, int nhead, int ffn);
int opseq_encoder_layer_f32(float *z, const float *in_w, const float *in_b, const float *out_w,
                            const float *out_b, const float *l1_w, const float *l1_b, const float *l2_w,
                            const float *l2_b, const float *n1_w, const float *n1_b, const float *n2_w,
                            const float *n2_b, void *workspace, size_t workspace_bytes, long S, int E,
                            int nhead, int ffn, void *stream);
/* The same layer over n_seg INDEPENDENT sequences of S tokens each, z [n_seg * S, E] in place: the requests a server merges
 * into one pass (learned_models.py:176-197 called once per request by the reference).  The token-wise stages run over all
 * n_seg * S rows, attention stays inside a sequence, and every kernel is chosen from S alone - each sequence's rows are
 * bit-identical to opseq_encoder_layer_f32 on that sequence by itself.  workspace >= opseq_encoder_workspace_bytes(n_seg * S,
 * ...); n_seg * S * ffn * 4 must stay below 2 GiB (OPNET_ESHAPE beyond: split the requests over passes). */
int opseq_encoder_layer_segmented_f32(float *z, const float *in_w, const float *in_b, const float *out_w,
                                      const float *out_b, const float *l1_w, const float *l1_b, const float *l2_w,
                                      const float *l2_b, const float *n1_w, const float *n1_b, const float *n2_w,
                                      const float *n2_b, void *workspace, size_t workspace_bytes, long S, int n_seg,
                                      int E, int nhead, int ffn, void *stream);
/* the same with the token-wise products chosen by ALL n_seg * S rows (a served pass in its throughput form: large LDS-DMA tiles
 * instead of one sequence's K-split tiles); attention stays inside a sequence.  Each sequence agrees with its lone forward to
 * rounding (another summation order), not bit for bit. */
int opseq_encoder_layer_batched_f32(float *z, const float *in_w, const float *in_b, const float *out_w,
                                      const float *out_b, const float *l1_w, const float *l1_b, const float *l2_w,
                                      const float *l2_b, const float *n1_w, const float *n1_b, const float *n2_w,
                                      const float *n2_b, void *workspace, size_t workspace_bytes, long S, int n_seg,
                                      int E, int nhead, int ffn, void *stream);

/* The feed-forward block of that layer as ONE kernel (csrc/ffn_kernels.hip): y [M,E] = relu(x linear1^T + b1) linear2^T + b2
 * (learned_models.py:166-171 via nn.TransformerEncoderLayer.forward: linear2(dropout(relu(linear1(src)))), eval), the [M,ffn]
 * activations kept in LDS.  Bit-identical to the two token-wise products of opseq_encoder_layer_batched_f32, which calls it when
 * the shape fits (E == 256, ffn a multiple of 128, x below 2 GiB; OPSEQ_FFN_FUSED=0 keeps the two products).  No workspace. */
int opseq_ffn_fused_supported(long M, int E, int ffn);
/* host logic only: the tile plan of those kernels for M rows on `cus` compute units - workgroups 0 .. n_full - 1 own 64 rows each, the
 * other grid - n_full workgroups 16 * tail_frags rows each (what full rounds of 64-row tiles leave, cut evenly over the CUs) */
int opseq_ffn_fused_plan(long M, int cus, int *n_full, int *tail_frags, unsigned *grid);
int opseq_ffn_fused_f32(const float *x, const float *l1_w, const float *l1_b, const float *l2_w, const float *l2_b,
                        float *y, long M, int E, int ffn, void *stream);

/* ---- detector backbone primitives (torchvision fasterrcnn_resnet50_fpn built at
 *      object_detection/models.py:6-20, called at baselines/detector.py:71-86).  fp32, NHWC.  The
 *      network's arithmetic is torchvision 0.5.0's (not vendored): parity with the reference detector is
 *      UNPINNED; these are checked against oracle/detector_oracle.py (build-authored torch restatement).
 * conv: y[n,oy,ox,co] = act(sum x[n,oy*s-p+dy,ox*s-p+dx,ci] w[co][(dy*KW+dx)*Cin+ci] + bias[co] (+ residual)),
 *       w rows padded with zeros to KP (multiple of 16); Cin multiple of 4; frozen BatchNorm pre-folded. */
int opdet_conv2d_f32(const float *x, const float *w, const float *bias, const float *residual, float *y,
                     int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int KP,
                     int relu, void *stream);
/* The same conv with scratch for a K split: the deep, spatially small layers of ONE frame and the 1000-row FCs give too few 128-row
 * tiles to fill 256 CUs, so their K is walked by several workgroups per tile (partial sums in the workspace, added in slice order by a
 * second launch - deterministic).  _workspace_bytes: 0 when the shape is not split (the call then equals opdet_conv2d_f32 and
 * workspace may be null).  bias / residual / workspace 16-byte aligned. */
size_t opdet_conv2d_workspace_bytes(int N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int KP);
int opdet_conv2d_ws_f32(const float *x, const float *w, const float *bias, const float *residual, float *y, int N, int H, int W,
                        int Cin, int Cout, int KH, int KW, int stride, int pad, int KP, int relu, void *workspace,
                        size_t workspace_bytes, void *stream);

/* Winograd F(2 x 2, 3 x 3) form of a stride-1, pad-1 3 x 3 convolution (csrc/wino_kernels.hip; replaces the same torchvision conv2d call
 * as opdet_conv2d_f32 for the FPN output convs / RPN head conv / deep bottleneck conv2 of reference object_detection/models.py:6-20; cuDNN,
 * which the reference ran on, picks Winograd for these layers itself).  u: the transformed weights [16][Cout][Cin] made ONCE per weight set
 * by opdet_wino_weights_f32 from the packed weight w [Cout][KP] of opdet_conv2d_f32; workspace: opdet_conv2d_wino_workspace_bytes (the 16
 * position planes of the transformed input and of the products; capped by OPDET_WINO_WS_MB, larger passes run in chunks of images).
 * Cin % 16 == 0, Cout % 4 == 0; results agree with opdet_conv2d_f32 to ~2e-6 of max|y| (another summation order). */
size_t opdet_wino_weights_bytes(int Cin, int Cout);
int opdet_wino_weights_f32(const float *w, float *u, int Cin, int Cout, int KP, void *stream);
size_t opdet_conv2d_wino_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int opdet_conv2d_wino_f32(const float *x, const float *u, const float *bias, float *y, int N, int H, int W, int Cin, int Cout, int relu,
                          void *workspace, size_t workspace_bytes, void *stream);
/* FeaturePyramidNetwork's top-down step:  y = conv(x) + bias + nearest_upsample(top), top [N, TH, TW, Cout] (F.interpolate to the conv's
 * output size); the addition rides in the conv kernel's epilogue where the shape allows, else conv + in-place upsample_add - the same
 * fp32 operations in the same order.  workspace as for opdet_conv2d_ws_f32. */
int opdet_conv2d_up_f32(const float *x, const float *w, const float *bias, const float *top, float *y, int N, int H, int W,
                        int Cin, int Cout, int KH, int KW, int stride, int pad, int KP, int TH, int TW, void *workspace,
                        size_t workspace_bytes, void *stream);
/* conv3 and the downsample branch of a stage's first bottleneck as ONE product (torchvision Bottleneck.forward:
 * relu(bn3(conv3(out)) + downsample(x))): y = act(W[:, :Cin] . x + W[:, Cin:] . x2[every stride2-th pixel] + bias); both 1 x 1; x [N, H, W, Cin],
 * x2 [N, H2, W2, Cin2] with (H2 - 1) / stride2 + 1 == H, w [Cout][Cin + Cin2], bias = the two folded biases added.  Cin, Cin2 multiples of
 * 16, Cout of 4.  _workspace_bytes: scratch of the K split (0: none), -1 (and OPNET_ESHAPE from the call) when the shape is not one the
 * LDS-DMA kernel runs - the caller then runs the two convs (opdet_conv2d_ws_f32 with the residual). */
long long opdet_conv2d_dual_workspace_bytes(int N, int H, int W, int Cin, int H2, int W2, int Cin2, int stride2, int Cout);
int opdet_conv2d_dual_f32(const float *x, const float *x2, const float *w, const float *bias, float *y, int N, int H, int W,
                          int Cin, int H2, int W2, int Cin2, int stride2, int Cout, int relu, void *workspace,
                          size_t workspace_bytes, void *stream);
int opdet_maxpool3x3s2_f32(const float *x, float *y, int N, int H, int W, int C, void *stream);
int opdet_subsample2_f32(const float *x, float *y, int N, int H, int W, int C, void *stream);
int opdet_upsample_add_f32(const float *lateral, const float *top, float *y, int N, int H, int W, int C,
                           int TH, int TW, void *stream);
/* frame_bgr: DEVICE uint8 [H,W,3]; mean/std: HOST float[3] (RGB). BGR->RGB, /256 (detector.py:75-76), normalise,
 * bilinear resize (align_corners=False) to [RH,RW], zero-pad to [PH,PW], 4 channels (4th = 0). */
int opdet_preprocess_frame_f32(const unsigned char *frame_bgr, float *y, int H, int W, int RH, int RW, int PH,
                               int PW, const float *mean3_host, const float *std3_host, void *stream);

/* ---- transformer encoder layer, TRAINING (reference learned_models.py:166-168 under training_main.py:183-217) --------
 * forward keeps the layer's activations (incl. the nhead S x S softmax matrices) in `saved` for the backward call;
 * `scratch` is shared by all layers / both calls.  Dropout sites of nn.TransformerEncoderLayer (attention weights,
 * after out_proj, after ReLU, after linear2) use a counter-based generator keyed by (seed, site, index); p_drop = 0
 * disables them (the only configuration that can be pinned against the reference - torch's masks are not reproducible).
 * backward OVERWRITES dz_in and the 12 gradients (state_dict layouts).  E, ffn multiples of 16, E <= 512,
 * S x roundup(S,16) x 4 B < 2 GiB. */
size_t opseq_encoder_train_saved_bytes(long S, int E, int nhead, int ffn);
size_t opseq_encoder_train_scratch_bytes(long S, int E, int nhead, int ffn);
int opseq_encoder_layer_train_forward_f32(const float *z_in, float *z_out, const float *in_w, const float *in_b,
                                          const float *out_w, const float *out_b, const float *l1_w, const float *l1_b,
                                          const float *l2_w, const float *l2_b, const float *n1_w, const float *n1_b,
                                          const float *n2_w, const float *n2_b, void *saved, size_t saved_bytes,
                                          void *scratch, size_t scratch_bytes, long S, int E, int nhead, int ffn,
                                          float p_drop, unsigned long long seed, void *stream);
/* TEST-ONLY: layer calls made with `seed` take their dropout masks from these device buffers (one byte per element, nonzero =
 * keep; site 0 attention weights [nhead][S][S], 1 after out_proj [S][E], 2 after ReLU [S][ffn], 3 after linear2 [S][E]) instead of
 * the counter generator: the masks a reference training step drew (learned_models.py:166-168 at p = 0.1) are fed in this way
 * (tests/golden/transformer_dropout_train.npz).  slot 0..7; _clear empties the table.  Both synchronise the device. */
int opseq_encoder_test_masks_set(int slot, unsigned long long seed, const unsigned char *m0, const unsigned char *m1,
                                 const unsigned char *m2, const unsigned char *m3);
int opseq_encoder_test_masks_clear(void);
int opseq_encoder_layer_train_backward_f32(const float *dz_out, float *dz_in, const float *in_w, const float *out_w,
                                           const float *l1_w, const float *l2_w, const float *n1_w, const float *n2_w,
                                           float *g_in_w, float *g_in_b, float *g_out_w, float *g_out_b, float *g_l1_w,
                                           float *g_l1_b, float *g_l2_w, float *g_l2_b, float *g_n1_w, float *g_n1_b,
                                           float *g_n2_w, float *g_n2_b, const void *saved, size_t saved_bytes,
                                           void *scratch, size_t scratch_bytes, long S, int E, int nhead, int ffn,
                                           float p_drop, unsigned long long seed, void *stream);

/* ---- detector back half: RPN proposals, MultiScaleRoIAlign, detections (the non-conv stages of the
 *      torchvision fasterrcnn_resnet50_fpn call at reference detector.py:84; PARITY UNPINNED).
 * The dense stages (RPNHead convs, TwoMLPHead, FastRCNNPredictor) run on opdet_conv2d_f32.
 *
 * opdet_rpn_proposals_f32: head_out = HOST array of n_levels DEVICE pointers, level l = [gh[l], gw[l], 16] fp32 NHWC
 * with channels 0..2 the objectness logits of the 3 anchors, 3..14 their deltas (anchor-major dx,dy,dw,dh), 15 unused.
 * AnchorGenerator(sizes = anchor_sizes[l], ratios 0.5/1/2, strides int(padded/grid)), per-level top pre_nms_top_n on
 * the logits, decode (weights 1), clip to [image_h, image_w], drop sides < min_size, per-level NMS, best
 * post_nms_top_n.  Out: proposals [post_nms_top_n, 4] xyxy (rows >= *count zero), scores (may be NULL), count
 * (DEVICE int).  Nothing is copied to the host; gh / gw / anchor_sizes are HOST arrays.
 *
 * The *_batch_* forms serve the n_images (1..64) equally sized images of one pass with ONE launch per stage (detector.py:84 is one
 * call per frame; the frames of a video share a size): every device array gains a leading image dimension - head_out level l =
 * [n_images, gh, gw, 16], proposals [n_images, post_nms_top_n, 4], scores [n_images, post], count [n_images]; feats level l =
 * [n_images, fh, fw, C], rois [n_images, max_rois, 4], out [n_images, max_rois, 7, 7, C]; class_logits [n_images, max_rois, NC], ...,
 * boxes [n_images, max_det, 4], n_det [n_images] - and the workspace is n_images blocks of the one-image size.  logits_stride /
 * reg_stride: floats between consecutive rois' rows of class_logits / box_regression (0 = dense: NC / 4 NC) - the two predictors
 * run as ONE product whose output row holds both.  Each image's result
 * is bit-identical to its own one-image call (which is the same launch with one image). */
size_t opdet_rpn_workspace_bytes_batch(int n_images, int n_levels, const int *gh, const int *gw, const int *anchor_sizes, int padded_h,
                                       int padded_w, int pre_nms_top_n);
int opdet_rpn_proposals_batch_f32(const float *const *head_out, int n_images, int n_levels, const int *gh, const int *gw,
                                  const int *anchor_sizes, int image_h, int image_w, int padded_h, int padded_w,
                                  int pre_nms_top_n, int post_nms_top_n, float nms_thresh, float min_size, float *proposals,
                                  float *scores, int *count, void *workspace, size_t workspace_bytes, void *stream);
int opdet_roi_align_batch_f32(const float *const *feats, int n_images, const int *fh, const int *fw, int C, int image_h,
                              const float *rois, const int *count, int max_rois, float *out, void *stream);
size_t opdet_detections_workspace_bytes_batch(int n_images, int max_rois, int num_classes);
int opdet_detections_batch_f32(const float *class_logits, const float *box_regression, const float *proposals,
                               const int *count, int n_images, int max_rois, int num_classes, int logits_stride, int reg_stride,
                               int image_h, int image_w, int orig_h, int orig_w, float score_thresh, float nms_thresh, int max_det,
                               float *boxes, float *scores, long long *labels, int *n_det, void *workspace, size_t workspace_bytes,
                               void *stream);
size_t opdet_rpn_workspace_bytes(int n_levels, const int *gh, const int *gw, const int *anchor_sizes, int padded_h,
                                 int padded_w, int pre_nms_top_n);
int opdet_rpn_proposals_f32(const float *const *head_out, int n_levels, const int *gh, const int *gw,
                            const int *anchor_sizes, int image_h, int image_w, int padded_h, int padded_w,
                            int pre_nms_top_n, int post_nms_top_n, float nms_thresh, float min_size, float *proposals,
                            float *scores, int *count, void *workspace, size_t workspace_bytes, void *stream);
/* MultiScaleRoIAlign(["0".."3"], 7, sampling_ratio 2): feats = HOST array of 4 DEVICE pointers [fh[l], fw[l], C] NHWC,
 * rois [max_rois, 4] in resized-image pixels, count DEVICE int; out [max_rois, 7, 7, C] (rows >= *count zero). */
int opdet_roi_align_f32(const float *const *feats, const int *fh, const int *fw, int C, int image_h, const float *rois,
                        const int *count, int max_rois, float *out, void *stream);
/* RoIHeads.postprocess_detections + the transform's rescale to the original frame: softmax, per-class decode
 * (weights 10,10,5,5), clip, drop background, score > score_thresh (>= 0.05), sides >= 1e-2, per-class NMS, best
 * max_det.  class_logits [max_rois, NC], box_regression [max_rois, 4 NC]; out boxes [max_det, 4] xyxy in original-frame
 * pixels, scores [max_det] descending, labels [max_det] int64, n_det DEVICE int (rows >= *n_det zero). */
size_t opdet_detections_workspace_bytes(int max_rois, int num_classes);
int opdet_detections_f32(const float *class_logits, const float *box_regression, const float *proposals,
                         const int *count, int max_rois, int num_classes, int image_h, int image_w, int orig_h,
                         int orig_w, float score_thresh, float nms_thresh, int max_det, float *boxes, float *scores,
                         long long *labels, int *n_det, void *workspace, size_t workspace_bytes, void *stream);
/* TEST-ONLY: the detector's own stable radix sort of (key, value) pairs (csrc/det_sort_kernels.hip; it orders the RPN candidates and
 * the detections - the role torch.sort / topk plays in torchvision's filter_proposals / postprocess_detections), by itself, so a
 * test can hold it against torch.sort(stable=True).  Ascending by the low `bits` bits of the keys (8-byte keys when key64 != 0, else
 * 4-byte); both buffer pairs hold n elements and are overwritten; *result_in_out = 1: the sorted pairs are in (keys_out, vals_out),
 * 0: in (keys_in, vals_in) (an even number of passes). */
size_t opdet_test_sort_scratch_bytes(long n);
int opdet_test_sort_pairs(void *keys_in, unsigned *vals_in, void *keys_out, unsigned *vals_out, long n, int bits, int key64,
                          void *scratch, size_t scratch_bytes, int *result_in_out, void *stream);

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
