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
