"""CPU checks of the build-authored detector restatement (oracle/detector_oracle.py) - the yardstick of the HIP
detector stages - against independent brute-force statements and known torchvision constants, plus the host logic of
the perception driver.  (Parity with torchvision itself is UNPINNED: the library is absent)"""
import json
import os

import numpy as np
import pytest

from oracle import detector_oracle as do


def test_base_anchors_are_torchvisions():
    # AnchorGenerator((32,), (0.5, 1, 2)).generate_anchors: the well-known rounded cell anchors
    assert do.base_anchors(32).tolist() == [[-23, -11, 23, 11], [-16, -16, 16, 16], [-11, -23, 11, 23]]
    assert do.base_anchors(512).tolist() == [[-362, -181, 362, 181], [-256, -256, 256, 256], [-181, -362, 181, 362]]
    a = do.level_anchors(64, 2, 3, 16, 16)
    assert a.shape == (18, 4)
    assert a[0].tolist() == [-45, -23, 45, 23] and a[3].tolist() == [16 - 45, -23, 16 + 45, 23]      # x fastest, 3 per cell
    assert a[9].tolist() == [-45, 16 - 23, 45, 16 + 23]


def test_decode_and_clip():
    boxes = np.array([[10, 20, 50, 80], [0, 0, 4, 4]], np.float32)
    assert np.allclose(do.decode_boxes(np.zeros((2, 4), np.float32), boxes), boxes)
    d = np.array([[0.5, -0.25, np.log(2.0), 100.0]], np.float32)          # dh far above the clip log(1000/16)
    out = do.decode_boxes(d, boxes[:1])
    w, h = 40.0, 60.0
    assert out[0, 0] == pytest.approx(30 + 0.5 * w - w) and out[0, 2] == pytest.approx(30 + 0.5 * w + w)
    assert out[0, 3] - out[0, 1] == pytest.approx(h * 1000.0 / 16, rel=1e-5)
    assert do.clip_boxes(np.array([[-5, -5, 400, 300]], np.float32), (240, 320)).tolist() == [[0, 0, 320, 240]]


def _nms_brute(boxes, scores, thresh, groups=None):
    order = sorted(range(len(scores)), key=lambda i: (-scores[i], i))
    keep = []
    for i in order:
        ok = True
        for j in keep:
            if groups is not None and groups[i] != groups[j]:
                continue
            x1, y1 = max(boxes[i][0], boxes[j][0]), max(boxes[i][1], boxes[j][1])
            x2, y2 = min(boxes[i][2], boxes[j][2]), min(boxes[i][3], boxes[j][3])
            inter = np.float32(max(np.float32(x2 - x1), 0)) * np.float32(max(np.float32(y2 - y1), 0))
            ai = np.float32(boxes[i][2] - boxes[i][0]) * np.float32(boxes[i][3] - boxes[i][1])
            aj = np.float32(boxes[j][2] - boxes[j][0]) * np.float32(boxes[j][3] - boxes[j][1])
            if inter / (ai + aj - inter) > np.float32(thresh):
                ok = False
                break
        if ok:
            keep.append(i)
    return keep


@pytest.mark.parametrize("seed,grouped", [(0, False), (1, True), (2, True)])
def test_nms_matches_brute_force(seed, grouped):
    rng = np.random.default_rng(seed)
    n = 300
    xy = rng.uniform(0, 60, (n, 2)).astype(np.float32)
    wh = rng.uniform(5, 40, (n, 2)).astype(np.float32)
    boxes = np.concatenate([xy, xy + wh], axis=1)
    scores = rng.normal(size=n).astype(np.float32)
    scores[::7] = scores[3]                                  # ties: broken by index (stable sort)
    groups = rng.integers(0, 4, n) if grouped else None
    assert do.nms(boxes, scores, 0.5, groups).tolist() == _nms_brute(boxes, scores, 0.5, groups)


def _roi_align_scalar(feat, roi, scale, out=7, sr=2):
    H, W, C = feat.shape
    x0, y0, x1, y1 = (np.float32(v) * np.float32(scale) for v in roi)
    rw, rh = max(x1 - x0, np.float32(1)), max(y1 - y0, np.float32(1))
    bw, bh = rw / np.float32(out), rh / np.float32(out)
    res = np.zeros((out, out, C), np.float64)
    for ph in range(out):
        for pw in range(out):
            for iy in range(sr):
                for ix in range(sr):
                    y = y0 + ph * bh + (iy + 0.5) * bh / sr
                    x = x0 + pw * bw + (ix + 0.5) * bw / sr
                    if y < -1 or y > H or x < -1 or x > W:
                        continue
                    y, x = max(y, 0), max(x, 0)
                    yl, xl = int(y), int(x)
                    if yl >= H - 1:
                        yl = yh = H - 1; y = yl
                    else:
                        yh = yl + 1
                    if xl >= W - 1:
                        xl = xh = W - 1; x = xl
                    else:
                        xh = xl + 1
                    ly, lx = y - yl, x - xl
                    res[ph, pw] += ((1 - ly) * (1 - lx) * feat[yl, xl] + (1 - ly) * lx * feat[yl, xh]
                                    + ly * (1 - lx) * feat[yh, xl] + ly * lx * feat[yh, xh])
    return res / (sr * sr)


def test_roi_align_matches_scalar_statement():
    rng = np.random.default_rng(0)
    feat = rng.normal(size=(9, 11, 3)).astype(np.float32)
    rois = np.array([[4, 6, 30, 28], [-8, -8, 10, 10], [20, 10, 200, 90], [38, 30, 39, 30.5], [-40, -40, -30, -30]], np.float32)
    got = do.roi_align(feat, rois, 0.25)
    for r, g in zip(rois, got):
        assert np.abs(g - _roi_align_scalar(feat, r, 0.25)).max() < 1e-5
    assert np.abs(got[4]).max() == 0.0                       # entirely outside: every sample is dropped


def test_level_mapper():
    side = np.array([10, 111, 113, 223, 225, 447, 449, 2000], np.float32)
    boxes = np.stack([np.zeros_like(side), np.zeros_like(side), side, side], axis=1)
    assert do.map_levels(boxes).tolist() == [0, 0, 1, 1, 2, 2, 3, 3]     # floor(4 + log2(s/224)) clamped to 2..5


def test_proposals_and_detections_invariants():
    rng = np.random.default_rng(3)
    outs = [rng.normal(0, 1.0, size=(h, w, 16)).astype(np.float32) for h, w in ((12, 18), (6, 9), (3, 5))]
    for o in outs:
        o[..., 3:15] *= 0.3
    b, s, l = do.rpn_proposals(outs, (90, 140), (96, 144), pre_nms_top_n=100, post_nms_top_n=60)
    assert len(b) == 60 and np.all(np.diff(s) <= 0)
    assert b.min() >= 0 and b[:, [0, 2]].max() <= 140 and b[:, [1, 3]].max() <= 90
    # within a level no surviving pair overlaps by more than the threshold
    for lv in set(l.tolist()):
        bb = b[l == lv]
        for i in range(len(bb)):
            for j in range(i + 1, len(bb)):
                iw = max(min(bb[i, 2], bb[j, 2]) - max(bb[i, 0], bb[j, 0]), 0)
                ih = max(min(bb[i, 3], bb[j, 3]) - max(bb[i, 1], bb[j, 1]), 0)
                a = (bb[i, 2] - bb[i, 0]) * (bb[i, 3] - bb[i, 1]) + (bb[j, 2] - bb[j, 0]) * (bb[j, 3] - bb[j, 1]) - iw * ih
                assert iw * ih / a <= 0.7 + 1e-6
    logits = rng.normal(0, 3.0, size=(60, do.NUM_CLASSES)).astype(np.float32)
    reg = rng.normal(0, 1.0, size=(60, 4 * do.NUM_CLASSES)).astype(np.float32)
    det = do.postprocess_detections(logits, reg, b, (90, 140), (60, 80))
    assert len(det["scores"]) <= 100 and np.all(np.diff(det["scores"]) <= 0) and det["scores"].min() > 0.05
    assert det["labels"].min() >= 1 and det["boxes"][:, [0, 2]].max() <= 80 + 1e-4 and det["boxes"][:, [1, 3]].max() <= 60 + 1e-4


def test_perception_driver_host_logic(tmp_path):
    from objectpermanence_amd.preprocess_perception_main import get_experiment_videos, read_video_frames
    frames = np.arange(2 * 4 * 5 * 3, dtype=np.uint8).reshape(2, 4, 5, 3)
    np.save(tmp_path / "b.npy", frames)
    np.savez(tmp_path / "a.npz", frames=frames)
    open(tmp_path / "notes.txt", "w").write("x")
    cfg = {"videos_dir": str(tmp_path)}
    assert [os.path.basename(p) for p in get_experiment_videos(cfg)] == ["a.npz", "b.npy"]
    open(tmp_path / "sample.txt", "w").write("/elsewhere/b.avi\n")
    assert [os.path.basename(p) for p in get_experiment_videos({**cfg, "sample_file": str(tmp_path / "sample.txt")})] == ["b.npy"]
    for src in (tmp_path / "b.npy", tmp_path / "a.npz", frames, list(frames)):
        got = list(read_video_frames(src))
        assert len(got) == 2 and np.array_equal(np.stack(got), frames)
    with pytest.raises(RuntimeError):
        list(read_video_frames(tmp_path / "clip.avi"))          # needs cv2, which this image does not have
