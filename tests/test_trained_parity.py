"""Non-vacuous mean-IoU / mAP@0.5 parity with TRAINED weights.

tests/golden/opnet_trained_fp16.npz: OPNet trained on the MI355X with this repo's own training path
(tools/train_synthetic.py, 40 epochs, synthetic clips), stored rounded to fp16.
tests/golden/opnet_trained_eval.npz: the REFERENCE model + ResultsAnalyzer on 16 held-out clips with
those weights (oracle/gen_golden.py): mean-IoU 0.599, mAP@0.5 0.793."""
import os

import numpy as np
import pytest

from oracle import opnet_oracle as oo, synth

CFG = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}


def _load(golden_dir):
    w = np.load(os.path.join(golden_dir, "opnet_trained_fp16.npz"))
    g = np.load(os.path.join(golden_dir, "opnet_trained_eval.npz"))
    params = {k: w[k].astype(np.float32) for k in w.files}
    boxes, labels = synth.make_batch(int(g["first"]), int(g["n"]), 300)
    return params, g, boxes, labels


def test_oracle_with_trained_weights(golden_dir):
    from oracle import c_oracle
    params, g, boxes, labels = _load(golden_dir)
    y, _ = c_oracle.opnet_forward(boxes, params)
    assert np.abs(y - g["y"]).max() < 2e-5
    px = oo.postprocess_to_pixels(g["y"])
    assert np.array_equal(px, g["pred_px"])
    miou, map50 = oo.mean_iou_and_map(px, oo.postprocess_to_pixels(labels))
    assert miou == pytest.approx(float(g["video_mean_iou"].mean()), abs=1e-12)
    assert map50 == pytest.approx(float(g["video_map50"].mean()), abs=1e-12)
    assert miou > 0.5 and map50 > 0.7          # the fixture is a model that actually tracks the snitch


@pytest.mark.gpu
def test_hip_mean_iou_matches_reference_with_trained_weights(golden_dir):
    import torch
    from objectpermanence_amd import ModelsFactory, metrics
    params, g, boxes, labels = _load(golden_dir)
    m = ModelsFactory.get_model("opnet", CFG)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    m.eval().to("cuda:0")
    with torch.no_grad():
        y, _ = m(torch.from_numpy(boxes).cuda())
    pred_px, gt_px, iou = metrics.postprocess_and_iou(y, torch.from_numpy(labels).cuda())
    torch.cuda.synchronize()
    assert np.abs(y.cpu().numpy() - g["y"]).max() < 2e-5
    flips = (pred_px.cpu().numpy() != g["pred_px"])
    assert flips.mean() < 2e-3 and np.abs(pred_px.cpu().numpy() - g["pred_px"]).max() <= 1
    miou, map50 = metrics.mean_iou_and_map(iou)
    assert miou == pytest.approx(float(g["video_mean_iou"].mean()), abs=1e-3)      # north_star bar
    assert map50 == pytest.approx(float(g["video_map50"].mean()), abs=2e-3)
    ref_iou = np.stack([oo.iou_for_video(p, q) for p, q in zip(g["pred_px"], gt_px.cpu().numpy())])
    assert (np.abs(iou.cpu().numpy() - ref_iou) > 1e-4).mean() < 5e-3             # only the .0-boundary frames
    per_video = np.abs(iou.cpu().numpy().mean(axis=1) - g["video_mean_iou"])
    assert per_video.max() < 1e-3
