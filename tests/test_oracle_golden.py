"""Pin the CPU oracle (oracle/opnet_oracle.py) against outputs of the reference itself.

The fixtures in tests/golden/ were produced by oracle/gen_golden.py, which imports
/root/reference/baselines/learned_models.py (OPNet) and tracking_utils.py (ResultsAnalyzer).
"""
import json
import os

import numpy as np
import pytest

from oracle import opnet_oracle as oo
from oracle import synth


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


@pytest.mark.parametrize("tag,dtype,tol_y,tol_l", [
    ("tiny", np.float64, 2e-6, 2e-5),
    ("tiny", np.float32, 5e-6, 5e-5),
    ("real", np.float64, 1e-5, 5e-5),
    ("real", np.float32, 1e-5, 5e-5),
])
def test_opnet_forward_matches_reference(golden_dir, tag, dtype, tol_y, tol_l):
    g = _load(golden_dir, f"opnet_{tag}.npz")
    cfg = json.loads(str(g["cfg"]))
    n, t = int(g["n_clips"]), int(g["t_frames"])
    boxes, _ = synth.make_batch(0, n, t)
    params = synth.opnet_synth_params(cfg)
    y, logits = oo.opnet_forward(boxes, params, dtype=dtype)
    assert y.shape == g["y"].shape and logits.shape == g["logits"].shape
    # error grows with t - check the last frames explicitly
    assert np.abs(y - g["y"]).max() < tol_y
    assert np.abs(y[:, -5:] - g["y"][:, -5:]).max() < tol_y
    assert np.abs(logits - g["logits"]).max() < tol_l
    # outputs must be non-degenerate or the test is vacuous
    assert g["y"].std() > 0.2


def test_opnet_intermediates_tiny(golden_dir):
    g = _load(golden_dir, "opnet_tiny.npz")
    cfg = json.loads(str(g["cfg"]))
    boxes, _ = synth.make_batch(0, int(g["n_clips"]), int(g["t_frames"]))
    _, _, im = oo.opnet_forward(boxes, synth.opnet_synth_params(cfg), np.float64, True)
    for k in ("h1", "probs", "frames_boxes", "h2"):
        assert np.abs(im[k] - g[k]).max() < 2e-6, k


def test_clips_are_independent(golden_dir):
    """batch-vs-single output identical -> clips shard across GPUs freely."""
    g = _load(golden_dir, "opnet_real.npz")
    assert np.array_equal(g["y_clip0_alone"][0], g["y"][0]) or \
        np.abs(g["y_clip0_alone"][0] - g["y"][0]).max() < 1e-6
    cfg = json.loads(str(g["cfg"]))
    boxes, _ = synth.make_batch(0, 1)
    y, _ = oo.opnet_forward(boxes, synth.opnet_synth_params(cfg), np.float32)
    assert np.abs(y[0] - g["y"][0]).max() < 1e-5


def test_postprocess_and_metric_match_results_analyzer(golden_dir):
    g = _load(golden_dir, "opnet_real.npz")
    m = _load(golden_dir, "metric.npz")
    _, labels = synth.make_batch(0, 4)
    # integer outputs: bit-exact
    assert np.array_equal(oo.postprocess_to_pixels(g["y"]), m["pred_px"])
    assert np.array_equal(oo.postprocess_to_pixels(labels), m["gt_px"])
    for tag in ("pred", "jit"):
        kept = m[f"kept_{tag}"]
        p = m[f"{tag}_px"][kept]
        gt = m["gt_px"][kept]
        ious = np.stack([oo.iou_for_video(a, b) for a, b in zip(p, gt)])
        assert np.array_equal(np.nan_to_num(ious, nan=-1.0), np.nan_to_num(m[f"iou_{tag}"], nan=-1.0))
        assert np.array_equal(ious.mean(axis=1), m[f"video_mean_iou_{tag}"], equal_nan=True)
        assert np.array_equal((ious > 0.5).mean(axis=1), m[f"video_map50_{tag}"])
    miou, map50 = oo.mean_iou_and_map(m["jit_px"], m["gt_px"])
    assert miou == pytest.approx(m["video_mean_iou_jit"].mean(), abs=1e-15)
    assert map50 == pytest.approx(m["video_map50_jit"].mean(), abs=1e-15)
    assert 0.2 < miou < 0.8  # non-vacuous


def test_synth_is_deterministic():
    a = synth.synth_tensor("x", (7, 5), 0.5)
    b = synth.synth_tensor("x", (7, 5), 0.5)
    assert np.array_equal(a, b) and a.dtype == np.float32 and np.abs(a).max() <= 0.5
    assert not np.array_equal(a, synth.synth_tensor("y", (7, 5), 0.5))
    boxes, labels = synth.make_clip(3)
    assert boxes.shape == (300, 15, 6) and labels.shape == (300, 4)
    # slot layout contract (datasets.py:265-336): invisible rows zero except a cone's cone bit
    vis = boxes[..., 4]
    assert set(np.unique(vis)) <= {0.0, 1.0}
    assert np.all(boxes[vis == 0][:, :5] == 0)
    assert np.all(boxes[:, 10:] == 0)
    assert np.all(boxes[:, 1:3, 5] == 1) and np.all(boxes[:, 0, 5] == 0)
