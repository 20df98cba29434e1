"""GPU parity tests that close the small holes VERDICT round 1 listed: the *_no_labels loss, the detector's frame
preparation, a full-size training step.  `pytest -m gpu`."""
import ctypes
import json
import os

import numpy as np
import pytest
import torch

from oracle import synth, torch_port

pytestmark = pytest.mark.gpu


def test_no_labels_loss_and_gradients_match_reference(golden_dir):
    """training_main.py:192-210 (masked L1 + 0.5 * consistency) through the reference's OPNet + autograd
    (tests/golden/opnet_no_labels_train.npz) vs training.compute_loss on the HIP model's output and the HIP backward"""
    from objectpermanence_amd import ModelsFactory
    from objectpermanence_amd.training import compute_loss
    g = np.load(os.path.join(golden_dir, "opnet_no_labels_train.npz"))
    cfg = json.loads(str(g["cfg"]))
    m = ModelsFactory.get_model("opnet", cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.opnet_synth_params(cfg).items()})
    m = m.to("cuda:0").train(True)
    boxes, labels = synth.make_batch(0, int(g["n_clips"]), int(g["t_frames"]))
    y, _ = m(torch.from_numpy(boxes).cuda())
    loss, pred, cons = compute_loss("baseline_lstm_no_labels", y, torch.from_numpy(labels).cuda(), torch.from_numpy(g["mask"]).cuda())
    loss.backward()
    assert float(loss) == pytest.approx(float(g["loss"]), abs=2e-6)
    assert float(pred) == pytest.approx(float(g["pred_loss"]), abs=2e-6)
    assert float(cons) == pytest.approx(float(g["consistency_loss"]), abs=2e-6)
    for k, p in m.named_parameters():
        ref = g["grad/" + k]
        assert np.abs(p.grad.cpu().numpy() - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), k


def test_detector_frame_preparation_matches_reference(golden_dir):
    """detector.py:74-80 run as written (tests/golden/detector_preprocess.npz: BGR -> RGB, / 256, float32, CHW) vs the HIP
    preprocess kernel with the later stages switched off (mean 0, std 1, no resize, no padding)"""
    from objectpermanence_amd import _lib
    lib = _lib.load()
    g = np.load(os.path.join(golden_dir, "detector_preprocess.npz"))
    frame, ref = g["frame"], g["tensor"]                       # [H,W,3] uint8 BGR, [1,3,H,W] float32
    h, w = frame.shape[:2]
    fr = torch.from_numpy(frame.copy()).cuda()
    y = torch.empty((h, w, 4), dtype=torch.float32, device="cuda:0")
    mean, std = (ctypes.c_float * 3)(0.0, 0.0, 0.0), (ctypes.c_float * 3)(1.0, 1.0, 1.0)
    rc = lib.opdet_preprocess_frame_f32(fr.data_ptr(), y.data_ptr(), h, w, h, w, h, w, mean, std,
                                        torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "opdet_preprocess_frame_f32")
    torch.cuda.synchronize()
    got = y.cpu().numpy()
    assert np.array_equal(got[..., :3].transpose(2, 0, 1)[None], ref)          # x / 256 is exact in fp32
    assert np.all(got[..., 3] == 0)


def test_full_size_training_step_matches_cpu_port():
    """BASELINE config 2 at full size (32 clips x 300 frames): loss and all six gradients of the HIP step against
    oracle/torch_port (the graph on torch's CPU ops, itself pinned by the reference-autograd goldens)"""
    from objectpermanence_amd import ModelsFactory, l1_mean
    cfg = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}
    params = synth.opnet_synth_params(cfg)
    boxes, labels = synth.make_batch(0, 32, 300)
    m = ModelsFactory.get_model("opnet", cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    m = m.to("cuda:0").train(True)
    y, _ = m(torch.from_numpy(boxes).cuda())
    loss = l1_mean(y, torch.from_numpy(labels).cuda())
    loss.backward()
    ref_loss, ref, _ = torch_port.loss_and_grads(boxes, labels, params)
    assert float(loss) == pytest.approx(float(ref_loss), rel=2e-5)
    for k, p in m.named_parameters():
        gmax = np.abs(ref[k]).max()
        assert np.abs(p.grad.cpu().numpy() - ref[k]).max() <= 5e-4 * gmax, k
