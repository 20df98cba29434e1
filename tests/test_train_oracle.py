"""Pin oracle/torch_port.py (gradients, L1 loss, Adam) against the reference's own training step
(fixtures from oracle/gen_golden.py: reference OPNet + torch autograd + torch.optim.Adam)."""
import json
import os

import numpy as np
import pytest

from oracle import synth, torch_port


def sample_indices(name, n, k=4096):
    if n <= k:
        return np.arange(n)
    u = synth.counter_uniform(synth.name_seed(name, 99), k)
    return np.unique((u * n).astype(np.int64))


def test_tiny_training_steps_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "opnet_train_tiny.npz"))
    cfg = json.loads(str(g["cfg"]))
    boxes, labels = synth.make_batch(0, int(g["n_clips"]), int(g["t_frames"]))
    params = synth.opnet_synth_params(cfg)
    state = {}
    for step, ref_loss in enumerate(g["losses"]):
        loss, grads, _ = torch_port.loss_and_grads(boxes, labels, params)
        assert loss == pytest.approx(float(ref_loss), abs=2e-6)
        if step == 0:
            for k, gr in grads.items():
                ref = g["grad/" + k]
                assert np.abs(gr - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max()), k
        torch_port.adam_step(params, grads, state)
    for k, w in params.items():
        assert np.abs(w - g["w_after/" + k]).max() < 2e-5, k   # 3 Adam steps of 1e-3 each


def test_real_gradients_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "opnet_train_real.npz"))
    cfg = json.loads(str(g["cfg"]))
    boxes, labels = synth.make_batch(0, int(g["n_clips"]), int(g["t_frames"]))
    params = synth.opnet_synth_params(cfg)
    loss, grads, _ = torch_port.loss_and_grads(boxes, labels, params)
    assert loss == pytest.approx(float(g["losses"][0]), abs=2e-6)
    for k, gr in grads.items():
        idx = g["gidx/" + k]
        assert np.array_equal(idx, sample_indices(k, gr.size))
        ref = g["gval/" + k]
        scale = max(1e-3, np.abs(ref).max())
        assert np.abs(gr.reshape(-1)[idx] - ref).max() <= 2e-4 * scale, k
        assert np.sqrt((gr.astype(np.float64) ** 2).sum()) == pytest.approx(float(g["gnorm/" + k]), rel=1e-4)


def test_torch_lstm_port_matches_explicit_port():
    """OPNetTorch (torch's CPU LSTM op; the timed CPU baseline of bench.py --mode train) == the explicit restatement"""
    import torch
    from oracle import synth, torch_port
    cfg = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 32, "videos_hidden_dim": 48}
    p = synth.opnet_synth_params(cfg)
    boxes, labels = synth.make_batch(3, 3, 9)
    loss, grads, y = torch_port.loss_and_grads(boxes, labels, p)
    m = torch_port.OPNetTorch(p)
    y2 = m(torch.from_numpy(boxes))
    l2 = torch_port.l1_mean(y2, torch.from_numpy(labels))
    l2.backward()
    assert float(l2) == pytest.approx(loss, abs=1e-6) and np.abs(y2.detach().numpy() - y).max() < 1e-5
    assert np.abs(m.lstm2.weight_hh_l0.grad.numpy() - grads["video_LSTM.weight_hh_l0"]).max() < 1e-5
    assert np.abs(m.lstm1.weight_ih_l0.grad.numpy() - grads["object_to_track_LSTM.weight_ih_l0"]).max() < 1e-5
