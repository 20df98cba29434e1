"""CPU restatement (numpy) of the reference's OPNet hot path.

TEST INFRASTRUCTURE ONLY: this module is the parity checker. Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the
product package (objectpermanence_amd/) never does and fails loudly when its
HIP library is missing.

Parity pin: the reference has no tests or golden vectors of its own
(SURVEY.md section 4), and its arithmetic lives in a third-party dependency that
is not under /root/reference: torch==1.4.0 (reference environment.yml:97) -
nn.LSTM, nn.Linear, F.softmax, torch.einsum. This file restates their
published algorithms; it is pinned against outputs of the reference itself,
generated in the build container by oracle/gen_golden.py (which imports
/root/reference/baselines/learned_models.py under torch 2.10 CPU) and
committed under tests/golden/. tests/test_oracle_golden.py is the pin.

Every function cites the reference line it follows (paths are relative to
/root/reference).
"""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np

FRAME_SHAPES = np.array([320, 240, 320, 240])  # baselines/inference_main.py:192


def _sigmoid(x: np.ndarray) -> np.ndarray:
    return 1.0 / (1.0 + np.exp(-x))


def lstm_seq(x: np.ndarray, w_ih: np.ndarray, w_hh: np.ndarray,
             h0: np.ndarray = None, c0: np.ndarray = None,
             return_state: bool = False):
    """Single-layer, unidirectional, batch_first, bias-free LSTM.

    Restates torch.nn.LSTM as constructed at baselines/learned_models.py:29,32
    (num_layers=1, bidirectional=False, batch_first=True, bias=False) and called at
    :39,:46 with zero initial state. Gate rows of W are chunked i, f, g, o:
        g_t = x_t W_ih^T + h_{t-1} W_hh^T
        c_t = sigmoid(f) c_{t-1} + sigmoid(i) tanh(g) ;  h_t = sigmoid(o) tanh(c_t)
    x [B,T,I], w_ih [4H,I], w_hh [4H,H] -> h [B,T,H].
    """
    B, T, _ = x.shape
    H = w_hh.shape[1]
    dt = x.dtype
    h = np.zeros((B, H), dtype=dt) if h0 is None else h0.astype(dt).copy()
    c = np.zeros((B, H), dtype=dt) if c0 is None else c0.astype(dt).copy()
    out = np.empty((B, T, H), dtype=dt)
    w_ih_t = np.ascontiguousarray(w_ih.T.astype(dt))
    w_hh_t = np.ascontiguousarray(w_hh.T.astype(dt))
    gx = x.reshape(B * T, -1) @ w_ih_t  # the input projection has no recurrence
    gx = gx.reshape(B, T, 4 * H)
    for t in range(T):
        g = gx[:, t] + h @ w_hh_t
        i = _sigmoid(g[:, 0 * H:1 * H])
        f = _sigmoid(g[:, 1 * H:2 * H])
        gg = np.tanh(g[:, 2 * H:3 * H])
        o = _sigmoid(g[:, 3 * H:4 * H])
        c = f * c + i * gg
        h = o * np.tanh(c)
        out[:, t] = h
    if return_state:
        return out, (h, c)
    return out


def softmax_lastdim(z: np.ndarray) -> np.ndarray:
    """F.softmax(dim=-1) (baselines/learned_models.py:41): max-subtracted exp / sum."""
    m = z.max(axis=-1, keepdims=True)
    e = np.exp(z - m)
    return e / e.sum(axis=-1, keepdims=True)


def opnet_forward(boxes: np.ndarray, p: Dict[str, np.ndarray], dtype=np.float64,
                  return_intermediates: bool = False):
    """OPNet.forward (baselines/learned_models.py:35-52).

    boxes [B,T,15,6] -> (y_boxes [B,T,4], logits [B,15,T]).
    p: state_dict names of learned_models.py:29-33.
    """
    B, T, S, F = boxes.shape
    x = boxes.astype(dtype)
    P = {k: v.astype(dtype) for k, v in p.items()}
    scene = x.reshape(B, T, S * F)                                   # :36-37 view
    h1 = lstm_seq(scene, P["object_to_track_LSTM.weight_ih_l0"],
                  P["object_to_track_LSTM.weight_hh_l0"])            # :39
    logits = h1 @ P["object_to_track_prediction.weight"].T          # :40
    probs = softmax_lastdim(logits)                                  # :41
    frames_boxes = np.einsum("bfot,bfo->bft", x, probs)              # :43
    h2 = lstm_seq(frames_boxes, P["video_LSTM.weight_ih_l0"],
                  P["video_LSTM.weight_hh_l0"])                      # :46
    y = h2 @ P["prediction_layer.weight"].T                          # :47
    logits_bct = np.ascontiguousarray(np.transpose(logits, (0, 2, 1)))  # :50 permute(0,2,1).contiguous()
    if return_intermediates:
        return y, logits_bct, {"h1": h1, "probs": probs, "frames_boxes": frames_boxes, "h2": h2}
    return y, logits_bct


# --------------------------------------------------------------------------------------
# output post-processing and the metric (integer arithmetic - bit-exact bar)
# --------------------------------------------------------------------------------------

def postprocess_to_pixels(y: np.ndarray) -> np.ndarray:
    """float32 normalised boxes -> int32 pixel boxes.

    baselines/inference_main.py:219 / training_main.py:97:
    ``(np.array(preds_float32) * frame_shapes_int64).astype(np.int32)`` - the multiply is carried
    out in float64 (float32 array x int64 array promotes to float64), the cast truncates toward 0.
    """
    y32 = np.asarray(y, dtype=np.float32)
    return (y32 * FRAME_SHAPES).astype(np.int32)


def iou_for_video(boxes_1: np.ndarray, boxes_2: np.ndarray) -> np.ndarray:
    """ResultsAnalyzer.compute_vectorized_iou_for_video (baselines/tracking_utils.py:137-159):
    integer pixel IoU with the inclusive +1 width/height convention. [T,4] x [T,4] -> [T] f64."""
    b1 = np.asarray(boxes_1)
    b2 = np.asarray(boxes_2)
    x11, y11, x12, y12 = (b1[:, k] for k in range(4))
    x21, y21, x22, y22 = (b2[:, k] for k in range(4))
    xa = np.maximum(x11, x21)
    ya = np.maximum(y11, y21)
    xb = np.minimum(x12, x22)
    yb = np.minimum(y12, y22)
    inter = np.maximum(xb - xa + 1, 0) * np.maximum(yb - ya + 1, 0)
    a1 = (x12 - x11 + 1) * (y12 - y11 + 1)
    a2 = (x22 - x21 + 1) * (y22 - y21 + 1)
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / (a1 + a2 - inter)


def mean_iou_and_map(pred_px: np.ndarray, gt_px: np.ndarray, thr: float = 0.5) -> Tuple[float, float]:
    """Dataset mean-IoU and mAP@thr as the reference aggregates them:
    per video mean over frames (tracking_utils.py:278-288 with np.mean), then the mean over videos
    (training_main.py:105-106); mAP uses the strict ``iou > thr`` (tracking_utils.py:251-256)."""
    ious = np.stack([iou_for_video(p, g) for p, g in zip(pred_px, gt_px)])
    video_mean = ious.mean(axis=1)
    video_map = (ious > thr).mean(axis=1)
    return float(video_mean.mean()), float(video_map.mean())


# --------------------------------------------------------------------------------------
# sibling reasoners (reference baselines/learned_models.py:55-197)
# --------------------------------------------------------------------------------------

def lstm_stack(x: np.ndarray, p: Dict[str, np.ndarray], prefix: str, num_layers: int) -> np.ndarray:
    """nn.LSTM(num_layers=n, bias=False, batch_first=True): layer l consumes layer l-1's outputs
    (learned_models.py:135-136, 170-171; no inter-layer dropout is configured)."""
    h = x
    for layer in range(num_layers):
        h = lstm_seq(h, p[f"{prefix}.weight_ih_l{layer}"], p[f"{prefix}.weight_hh_l{layer}"])
    return h


def baseline_lstm_forward(x: np.ndarray, p: Dict[str, np.ndarray], dtype=np.float64) -> np.ndarray:
    """BaselineLstm.forward (learned_models.py:104-118): [B,T,15,5] -> view [B,T,75] -> LSTM -> Linear."""
    B, T = x.shape[:2]
    P = {k: v.astype(dtype) for k, v in p.items()}
    h = lstm_seq(x.astype(dtype).reshape(B, T, -1), P["video_LSTM.weight_ih_l0"], P["video_LSTM.weight_hh_l0"])
    return h @ P["predictions_layer.weight"].T


def non_linear_lstm_forward(x: np.ndarray, p: Dict[str, np.ndarray], dtype=np.float64) -> np.ndarray:
    """NonLinearLstm.forward (learned_models.py:134-151): relu(Linear 5->F) per slot -> [B,T,15F] ->
    2-layer LSTM -> Linear."""
    B, T = x.shape[:2]
    P = {k: v.astype(dtype) for k, v in p.items()}
    feats = np.maximum(x.astype(dtype) @ P["boxes_linear.weight"].T, 0.0)      # :138
    h = lstm_stack(feats.reshape(B, T, -1), P, "video_LSTM", 2)                # :142-145
    return h @ P["predictions_layer.weight"].T                                 # :148


def opnet_lstm_mlp_forward(boxes: np.ndarray, p: Dict[str, np.ndarray], dtype=np.float64):
    """OPNetLstmMlp.forward (learned_models.py:72-89): OPNet with the video LSTM replaced by
    relu(Linear 6->H2) (:83)."""
    B, T, S, F = boxes.shape
    x = boxes.astype(dtype)
    P = {k: v.astype(dtype) for k, v in p.items()}
    h1 = lstm_seq(x.reshape(B, T, S * F), P["object_to_track_LSTM.weight_ih_l0"], P["object_to_track_LSTM.weight_hh_l0"])
    logits = h1 @ P["object_to_track_prediction.weight"].T
    probs = softmax_lastdim(logits)
    fb = np.einsum("bfot,bfo->bft", x, probs)
    hidden = np.maximum(fb @ P["hidden_layer.weight"].T, 0.0)
    y = hidden @ P["prediction_layer.weight"].T
    return y, np.ascontiguousarray(np.transpose(logits, (0, 2, 1)))


def layer_norm(x: np.ndarray, g: np.ndarray, b: np.ndarray, eps: float = 1e-5) -> np.ndarray:
    """nn.LayerNorm over the last dim: biased variance, eps inside the sqrt."""
    mu = x.mean(axis=-1, keepdims=True)
    var = ((x - mu) ** 2).mean(axis=-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * g + b


def encoder_layer(z: np.ndarray, P: Dict[str, np.ndarray], pre: str, nhead: int) -> np.ndarray:
    """One post-LN nn.TransformerEncoderLayer in eval mode (dropout off), ReLU FFN, on ONE sequence
    z [S, E] (torch.nn.MultiheadAttention: q scaled by 1/sqrt(head_dim), softmax over keys, no mask)."""
    S, E = z.shape
    hd = E // nhead
    qkv = z @ P[pre + "self_attn.in_proj_weight"].T + P[pre + "self_attn.in_proj_bias"]
    q, k, v = qkv[:, :E], qkv[:, E:2 * E], qkv[:, 2 * E:]
    heads = []
    for h in range(nhead):
        sl = slice(h * hd, (h + 1) * hd)
        sc = (q[:, sl] / np.sqrt(hd)) @ k[:, sl].T
        heads.append(softmax_lastdim(sc) @ v[:, sl])
    a = np.concatenate(heads, axis=1) @ P[pre + "self_attn.out_proj.weight"].T + P[pre + "self_attn.out_proj.bias"]
    z = layer_norm(z + a, P[pre + "norm1.weight"], P[pre + "norm1.bias"])
    f = np.maximum(z @ P[pre + "linear1.weight"].T + P[pre + "linear1.bias"], 0.0)
    f = f @ P[pre + "linear2.weight"].T + P[pre + "linear2.bias"]
    return layer_norm(z + f, P[pre + "norm2.weight"], P[pre + "norm2.bias"])


def transformer_lstm_forward(x: np.ndarray, p: Dict[str, np.ndarray], cfg: Dict[str, int], dtype=np.float64) -> np.ndarray:
    """TransformerLstm.forward (learned_models.py:174-197), eval mode, restated on its LIVE path.

    The reference feeds [B*T, 15, E] to a sequence-first encoder, so attention runs over the
    S = B*T frame axis independently per object slot (the "batch" axis of the encoder), across all
    clips of the minibatch and non-causally; only slot 0 is kept (:185).  Slots 1..14 never influence
    the output, so this restatement evaluates slot 0 only (tests pin it against the
    reference's full 15-slot evaluation, including the batch-composition dependence)."""
    B, T = x.shape[:2]
    P = {k: v.astype(dtype) for k, v in p.items()}
    z = np.maximum(x[:, :, 0, :].astype(dtype) @ P["boxes_linear.weight"].T, 0.0).reshape(B * T, -1)  # :178, slot 0
    for layer in range(cfg["num_attention_layers"]):
        z = encoder_layer(z, P, f"attention_encoder.layers.{layer}.", cfg["num_attention_heads"])          # :184
    h = lstm_stack(z.reshape(B, T, -1), P, "video_LSTM", cfg["num_lstm_layers"])                           # :189-192
    return h @ P["predictions_layer.weight"].T                                                              # :195
