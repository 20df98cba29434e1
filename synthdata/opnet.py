"""Deterministic synthetic weights and CATER-shaped clips.

DATA ONLY - seeded inputs and weights, no model arithmetic: imported by tests/, tools/, __graft_entry__.smoke(),
bench.py and (re-exported as oracle.synth) by the oracle; never by the product package.

Two generators live here:

* ``counter_uniform`` - a counter-based RNG (splitmix64 finaliser over
  ``seed * GOLDEN + index``) so that full-size weight tensors never need to
  be committed: the golden generator (oracle/gen_golden.py, which imports the
  reference) and the GPU-box tests regenerate bit-identical fp32 weights from
  a tensor name.

* ``make_clip`` - one synthetic CATER clip in the on-the-wire tensor format the
  reference's dataset hands to the model:
  ``boxes float32 [300, 15, 6]`` = per slot ``[x1/320, y1/240, x2/320, y2/240,
  visible, is_cone]`` (reference baselines/datasets.py:265-336 - invisible
  object -> all-zero row, invisible cone -> ``[0,0,0,0,0,1]``, slots beyond the
  video's objects all-zero, slot 0 = snitch) and ``labels float32 [300, 4]`` =
  snitch ground-truth ``xyxy / [320,240,320,240]`` (datasets.py:33-45).
"""
from __future__ import annotations

import zlib
from typing import Dict, Tuple

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)
_GOLDEN = 0x9E3779B97F4A7C15

T_FRAMES = 300
MAX_OBJECTS = 15
FRAME_SHAPES = np.array([320, 240, 320, 240], dtype=np.float64)


def _splitmix64(z: np.ndarray) -> np.ndarray:
    z = z.astype(np.uint64, copy=True)
    with np.errstate(over="ignore"):
        z = (z + np.uint64(_GOLDEN)) & _MASK
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
        z = z ^ (z >> np.uint64(31))
    return z


def counter_uniform(seed: int, n: int) -> np.ndarray:
    """n doubles in [0, 1), element i a pure function of (seed, i)."""
    idx = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = (np.uint64(seed & 0xFFFFFFFFFFFFFFFF) * np.uint64(_GOLDEN)) & _MASK
        z = _splitmix64((idx + base) & _MASK)
    return (z >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def name_seed(name: str, salt: int = 0) -> int:
    return (zlib.crc32(name.encode()) << 8) ^ (salt & 0xFF)


def synth_tensor(name: str, shape: Tuple[int, ...], bound: float, salt: int = 0) -> np.ndarray:
    """fp32 tensor ~ U(-bound, bound), keyed by its state_dict name."""
    n = int(np.prod(shape))
    u = counter_uniform(name_seed(name, salt), n)
    return ((2.0 * u - 1.0) * bound).astype(np.float32).reshape(shape)


# --------------------------------------------------------------------------------------
# model parameter sets (names/shapes: SURVEY.md section 8-a1 and section 11, verified against the
# reference's state_dict() in oracle/gen_golden.py)
# --------------------------------------------------------------------------------------

def opnet_shapes(cfg: Dict[str, int]) -> Dict[str, Tuple[int, ...]]:
    h1 = cfg["object_to_track_hidden_dim"]
    h2 = cfg["videos_hidden_dim"]
    nsel = cfg["object_to_track_pred_dim"]
    return {
        "object_to_track_LSTM.weight_ih_l0": (4 * h1, 6 * 15),
        "object_to_track_LSTM.weight_hh_l0": (4 * h1, h1),
        "object_to_track_prediction.weight": (nsel, h1),
        "video_LSTM.weight_ih_l0": (4 * h2, 6),
        "video_LSTM.weight_hh_l0": (4 * h2, h2),
        "prediction_layer.weight": (4, h2),
    }


def opnet_synth_params(cfg: Dict[str, int], salt: int = 0) -> Dict[str, np.ndarray]:
    """'Trained-like' synthetic OPNet weights.

    torch's default U(-1/sqrt(H), 1/sqrt(H)) init gives outputs of +-0.006 ->
    int-pixel boxes all zero and a near-uniform selection softmax, i.e. vacuous parity tests. The
    gains below were picked (oracle fp64 run, 4 clips) so that y spans about (-1.1, 2.0) with
    std 0.84, the slot softmax is peaked (mean max-prob 0.43) and the recurrences are
    moderate-gain, non-chaotic (fp32-vs-fp64 drift 2e-6 on y over 300 steps)."""
    h1 = cfg["object_to_track_hidden_dim"]
    h2 = cfg["videos_hidden_dim"]
    bounds = {
        "object_to_track_LSTM.weight_ih_l0": 2.0 / np.sqrt(h1),
        "object_to_track_LSTM.weight_hh_l0": 2.0 / np.sqrt(h1),
        "object_to_track_prediction.weight": 30.0 / np.sqrt(h1),
        "video_LSTM.weight_ih_l0": 1.0,
        "video_LSTM.weight_hh_l0": 2.0 / np.sqrt(h2),
        "prediction_layer.weight": 8.0 / np.sqrt(h2),
    }
    return {name: synth_tensor(name, shape, float(bounds[name]), salt)
            for name, shape in opnet_shapes(cfg).items()}


def _lstm_params(prefix, in_dim, hidden, layers, gain=2.0, salt=0):
    out = {}
    for l in range(layers):
        k_in = in_dim if l == 0 else hidden
        out[f"{prefix}.weight_ih_l{l}"] = synth_tensor(f"{prefix}.weight_ih_l{l}", (4 * hidden, k_in), gain / np.sqrt(max(k_in, 16)), salt)
        out[f"{prefix}.weight_hh_l{l}"] = synth_tensor(f"{prefix}.weight_hh_l{l}", (4 * hidden, hidden), gain / np.sqrt(hidden), salt)
    return out


def baseline_lstm_synth_params(cfg, salt=0):
    """BaselineLstm state_dict."""
    h = cfg["videos_hidden_dim"]
    p = _lstm_params("video_LSTM", 75, h, 1, salt=salt)
    p["predictions_layer.weight"] = synth_tensor("predictions_layer.weight", (4, h), 8.0 / np.sqrt(h), salt)
    return p


def non_linear_lstm_synth_params(cfg, salt=0):
    f, h = cfg["boxes_features_dim"], cfg["videos_hidden_dim"]
    p = {"boxes_linear.weight": synth_tensor("boxes_linear.weight", (f, 5), 0.9, salt)}
    p.update(_lstm_params("video_LSTM", 15 * f, h, 2, salt=salt))
    p["predictions_layer.weight"] = synth_tensor("predictions_layer.weight", (4, h), 8.0 / np.sqrt(h), salt)
    return p


def opnet_lstm_mlp_synth_params(cfg, salt=0):
    h1, h2 = cfg["object_to_track_hidden_dim"], cfg["videos_hidden_dim"]
    full = opnet_synth_params(cfg, salt)
    p = {k: v for k, v in full.items() if k.startswith("object_to_track")}
    p["hidden_layer.weight"] = synth_tensor("hidden_layer.weight", (h2, 6), 1.0, salt)
    p["prediction_layer.weight"] = synth_tensor("prediction_layer.weight", (4, h2), 4.0 / np.sqrt(h2), salt)
    return p


def transformer_lstm_synth_params(cfg, ffn=2048, salt=0):
    """TransformerLstm state_dict; dim_feedforward = 2048 is the torch default the
    reference relies on (learned_models.py:166)."""
    e, h = cfg["boxes_features_dim"], cfg["lstm_hidden_dim"]
    p = {"boxes_linear.weight": synth_tensor("boxes_linear.weight", (e, 5), 0.9, salt)}
    for l in range(cfg["num_attention_layers"]):
        pre = f"attention_encoder.layers.{l}."
        p[pre + "self_attn.in_proj_weight"] = synth_tensor(pre + "in_w", (3 * e, e), 2.0 / np.sqrt(e), salt)
        p[pre + "self_attn.in_proj_bias"] = synth_tensor(pre + "in_b", (3 * e,), 0.1, salt)
        p[pre + "self_attn.out_proj.weight"] = synth_tensor(pre + "out_w", (e, e), 1.5 / np.sqrt(e), salt)
        p[pre + "self_attn.out_proj.bias"] = synth_tensor(pre + "out_b", (e,), 0.05, salt)
        p[pre + "linear1.weight"] = synth_tensor(pre + "l1_w", (ffn, e), 1.5 / np.sqrt(e), salt)
        p[pre + "linear1.bias"] = synth_tensor(pre + "l1_b", (ffn,), 0.05, salt)
        p[pre + "linear2.weight"] = synth_tensor(pre + "l2_w", (e, ffn), 1.5 / np.sqrt(ffn), salt)
        p[pre + "linear2.bias"] = synth_tensor(pre + "l2_b", (e,), 0.05, salt)
        for n in ("norm1", "norm2"):
            p[pre + n + ".weight"] = (1.0 + synth_tensor(pre + n + "_w", (e,), 0.1, salt)).astype(np.float32)
            p[pre + n + ".bias"] = synth_tensor(pre + n + "_b", (e,), 0.05, salt)
    p.update(_lstm_params("video_LSTM", e, h, cfg["num_lstm_layers"], salt=salt))
    p["predictions_layer.weight"] = synth_tensor("predictions_layer.weight", (4, h), 8.0 / np.sqrt(h), salt)
    return p


def boxes5(boxes6: np.ndarray) -> np.ndarray:
    """the 5-track input of the non-OPNet models (datasets.py:128-196): the 6-track tensor without is_cone"""
    return np.ascontiguousarray(boxes6[..., :5])


# --------------------------------------------------------------------------------------
# synthetic clips
# --------------------------------------------------------------------------------------

SNITCH_ID = 140
# two cone ids and seven non-cone ids from the reference class table (object_indices.py:
# "*_cone_*" names are cones; ids 0 and 4 are large cones). Sorted snitch-first then ascending
# (datasets.py:47-54).
CONE_IDS = (0, 4)
OTHER_IDS = (65, 70, 98, 101, 133, 150, 171)


def make_clip(c: int, t_frames: int = T_FRAMES) -> Tuple[np.ndarray, np.ndarray]:
    """Synthetic clip c (seed 1000 + c). Returns (boxes [T,15,6] f32, labels [T,4] f32)."""
    rng = np.random.default_rng(1000 + c)
    ids = [SNITCH_ID] + sorted(CONE_IDS + OTHER_IDS)
    n_obj = len(ids)
    is_cone = np.array([1.0 if i in CONE_IDS else 0.0 for i in ids])

    # smooth integer random walks in pixels
    w = rng.integers(8, 65, size=n_obj)
    h = rng.integers(8, 65, size=n_obj)
    x1 = np.empty((t_frames, n_obj), dtype=np.int64)
    y1 = np.empty((t_frames, n_obj), dtype=np.int64)
    px = rng.uniform(0, 300 - 64, size=n_obj)
    py = rng.uniform(0, 220 - 64, size=n_obj)
    vx = rng.normal(0, 1.0, size=n_obj)
    vy = rng.normal(0, 1.0, size=n_obj)
    for t in range(t_frames):
        vx = 0.9 * vx + rng.normal(0, 0.6, size=n_obj)
        vy = 0.9 * vy + rng.normal(0, 0.6, size=n_obj)
        px = np.clip(px + vx, 0, 299 - 64)
        py = np.clip(py + vy, 0, 219 - 64)
        x1[t] = px.astype(np.int64)
        y1[t] = py.astype(np.int64)
    x2 = x1 + w[None, :]
    y2 = y1 + h[None, :]

    visible = rng.random((t_frames, n_obj)) < 0.9
    # the snitch disappears in 15-frame runs (containment / occlusion episodes)
    visible[:, 0] = True
    n_runs = int(rng.integers(2, 6))
    for _ in range(n_runs):
        s = int(rng.integers(5, max(6, t_frames - 20)))
        visible[s:s + 15, 0] = False

    boxes = np.zeros((t_frames, MAX_OBJECTS, 6), dtype=np.float64)
    raw = np.stack([x1, y1, x2, y2], axis=-1).astype(np.float64)  # [T, n, 4]
    norm = raw / FRAME_SHAPES
    vis_f = visible.astype(np.float64)
    boxes[:, :n_obj, :4] = norm * vis_f[..., None]
    boxes[:, :n_obj, 4] = vis_f
    boxes[:, :n_obj, 5] = is_cone[None, :]  # cone padding row keeps its cone bit (datasets.py:315-316)

    labels = raw[:, 0, :] / FRAME_SHAPES  # ground truth is known even when hidden
    return boxes.astype(np.float32), labels.astype(np.float32)


def make_batch(first_clip: int, n_clips: int, t_frames: int = T_FRAMES) -> Tuple[np.ndarray, np.ndarray]:
    bs, ls = zip(*(make_clip(first_clip + i, t_frames) for i in range(n_clips)))
    return np.stack(bs), np.stack(ls)


# --------------------------------------------------------------------------------------
# synthetic on-disk samples (formats of SURVEY.md section 10: <video>.pkl and <video>_bb.json)
# --------------------------------------------------------------------------------------
SNITCH_NAME = "small_gold_spl_metal_Spl_0"   # datasets.py:13


def make_raw_video(c: int, variant: str = "plain", t_frames: int = T_FRAMES):
    """Raw perception sample as the detector / "perfect perception" tools write it:
    returns (bb list[T] of int64 [n_f,4] xyxy pixels, labels list[T] of int64 [n_f], gt dict name -> list[T] [x,y,w,h]).
    variants: "plain" (10 objects, random drop-outs), "dups" (duplicate ids + shuffled order inside frames),
    "crowded" (17 distinct objects -> the encoder truncates to 15 slots), "nosnitch0" (snitch absent at frame 0),
    "sparse" (the two HIGHEST ids of the video are cones - 185, 189 -, every object is visible only half of the time and
    every 7th frame is empty: slots behind a frame's last detection must get plain zero padding, datasets.py:288-323)."""
    rng = np.random.default_rng(5000 + c)
    ids = [SNITCH_ID] + list(CONE_IDS) + list(OTHER_IDS)
    if variant == "crowded":
        ids = ids + [8, 12, 16, 66, 67, 99, 134]         # three more cones (8, 12, 16) and four non-cones
    if variant == "sparse":
        ids = ids[:-2] + [185, 189]                      # cones in the last two slots of the video's order
    n = len(ids)
    w = rng.integers(8, 65, size=n); h = rng.integers(8, 65, size=n)
    px = rng.uniform(0, 236, size=n); py = rng.uniform(0, 156, size=n)
    vx = rng.normal(0, 1, size=n); vy = rng.normal(0, 1, size=n)
    bbs, labels, gt = [], [], []
    hidden = np.zeros(t_frames, dtype=bool)
    for _ in range(int(rng.integers(2, 6))):
        s = int(rng.integers(1 if variant != "nosnitch0" else 0, max(2, t_frames - 20)))
        hidden[s:s + 15] = True
    if variant == "nosnitch0":
        hidden[:3] = True
    for t in range(t_frames):
        vx = 0.9 * vx + rng.normal(0, 0.6, size=n); vy = 0.9 * vy + rng.normal(0, 0.6, size=n)
        px = np.clip(px + vx, 0, 235); py = np.clip(py + vy, 0, 155)
        x1 = px.astype(np.int64); y1 = py.astype(np.int64)
        box = np.stack([x1, y1, x1 + w, y1 + h], axis=1)
        vis = rng.random(n) < (0.5 if variant == "sparse" else 0.9)
        vis[0] = not hidden[t]
        if variant == "sparse" and t % 7 == 3:
            vis[:] = False                                  # a frame without a single detection
        idx = np.flatnonzero(vis)
        if variant == "dups" and len(idx) > 2:
            extra = rng.choice(idx, size=2)                 # the perception model repeats two ids ...
            idx = np.concatenate([idx, extra])
            idx = idx[rng.permutation(len(idx))]            # ... and returns them in score order, not id order
        bbs.append(box[idx] + (rng.integers(-2, 3, size=(len(idx), 4)) if variant == "dups" else 0))
        labels.append(np.array([ids[i] for i in idx], dtype=np.int64))
        gt.append([int(x1[0]), int(y1[0]), int(w[0]), int(h[0])])
    return bbs, labels, {SNITCH_NAME: gt, "other_object_0": [[0, 0, 0, 0]] * t_frames}


def make_analysis_fixture(root: str, n_videos: int = 6, t_frames: int = T_FRAMES):
    """Synthetic analysis inputs in the reference's on-disk formats: prediction and label
    `<video>_bb.json` directories plus six frame-list TSVs.  Returns the argument dict of analyze_results()."""
    import json
    import os
    pred_dir, lab_dir = os.path.join(root, "pred"), os.path.join(root, "labels")
    os.makedirs(pred_dir, exist_ok=True); os.makedirs(lab_dir, exist_ok=True)
    names = [str(v) for v in (0, 1, 10, 11, 2, 3)][:n_videos]          # string sort differs from numeric order
    files = {k: [] for k in ("containment", "static", "move", "vis0", "vis30", "vis99")}
    for i, name in enumerate(names):
        rng = np.random.default_rng(9000 + i)
        _, _, gt = make_raw_video(i, "plain", t_frames)
        xywh = np.array(gt[SNITCH_NAME])
        xyxy = np.stack([xywh[:, 0], xywh[:, 1], xywh[:, 0] + xywh[:, 2], xywh[:, 1] + xywh[:, 3]], axis=1)
        pred = xyxy + rng.integers(-10, 11, size=xyxy.shape)
        json.dump(pred.tolist(), open(os.path.join(pred_dir, name + "_bb.json"), "w"))
        json.dump(gt, open(os.path.join(lab_dir, name + "_bb.json"), "w"))
        for j, k in enumerate(files):
            frames = [] if (i == 1 and k == "static") else sorted(set(int(x) for x in rng.integers(0, t_frames, size=15 * (j + 1) + 7 * i)))
            files[k].append(name + "\t" + ",".join(str(x) for x in frames) + "\n")
    paths = {}
    for k, lines in files.items():
        paths[k] = os.path.join(root, k + ".txt")
        open(paths[k], "w").writelines(lines)
    return dict(predictions_dir=pred_dir, labels_dir=lab_dir, containment_annotations=paths["containment"],
                containment_only_static=paths["static"], containment_with_movements=paths["move"],
                visibility_gt_0=paths["vis0"], visibility_gt_30=paths["vis30"], visibility_gt_99=paths["vis99"],
                iou_thresh=[0.5, 0.75])
