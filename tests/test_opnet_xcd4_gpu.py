"""GPU parity of the 4-clip per-XCD persistent kernels (csrc/opnet_xcd4_kernels.hip) - the form opnet_train_forward_f32 /
opnet_train_backward_f32 take for batches of up to 32 clips on a whole MI355X, and opnet_xcd4_forward_f32, the inference
forward of one small request - against the launch chain (OPNET_XCD4=0 / use_xcd4 = "0"), the reference's goldens, the
oracle, and the fp64 port of the reference (oracle/torch_port.py); tests/test_train_gpu.py runs on them by default."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import synth, torch_port

pytestmark = pytest.mark.gpu

REAL_CFG = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}


def _model():
    from objectpermanence_amd import ModelsFactory
    m = ModelsFactory.get_model("opnet", REAL_CFG)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.opnet_synth_params(REAL_CFG).items()})
    return m.to("cuda:0").train(True)


def _run(m, boxes, labels):
    from objectpermanence_amd import l1_mean
    m.zero_grad(set_to_none=True)
    y, logits = m(torch.from_numpy(boxes).cuda())
    loss = l1_mean(y, torch.from_numpy(labels).cuda())
    loss.backward()
    torch.cuda.synchronize()
    return (float(loss.item()), y.detach().cpu().numpy(), logits.detach().cpu().numpy(),
            {k: p.grad.cpu().numpy().copy() for k, p in m.named_parameters()})


def _supported():
    from objectpermanence_amd import _lib
    return bool(_lib.load().opnet_xcd_supported(256, 512))


@pytest.mark.parametrize("B,T", [(1, 1), (4, 7), (20, 9), (32, 12), (31, 40)])
def test_persistent_step_matches_chain_and_port(B, T, monkeypatch):
    if not _supported():
        pytest.skip("needs a whole MI355X (8 XCDs x 32 CUs)")
    boxes, labels = synth.make_batch(300 + B, B, T)
    m = _model()
    monkeypatch.setenv("OPNET_XCD4", "1")
    loss_x, y_x, lg_x, g_x = _run(m, boxes, labels)
    loss_x2, y_x2, lg_x2, g_x2 = _run(m, boxes, labels)
    monkeypatch.setenv("OPNET_XCD4", "0")
    loss_c, y_c, lg_c, g_c = _run(m, boxes, labels)
    assert np.isfinite(y_x).all()
    # run to run: the same bits
    assert np.array_equal(y_x, y_x2) and np.array_equal(lg_x, lg_x2)
    for k in g_x:
        assert np.array_equal(g_x[k], g_x2[k]), k
    # against the launch chain: another summation order of the same fp32 products
    assert np.abs(y_x - y_c).max() < 2e-5
    assert np.abs(lg_x - lg_c).max() < 5e-5
    assert loss_x == pytest.approx(loss_c, abs=2e-6)
    for k in g_x:
        assert np.abs(g_x[k] - g_c[k]).max() <= 2e-4 * max(1e-2, np.abs(g_c[k]).max()), k
    # against the fp64 port of the reference
    ref_loss, ref_grads, y_ref = torch_port.loss_and_grads(boxes, labels, synth.opnet_synth_params(REAL_CFG), dtype=torch.float64)
    assert np.abs(y_x - y_ref).max() < 2e-5
    assert loss_x == pytest.approx(ref_loss, abs=2e-6)
    for k, gr in g_x.items():
        assert np.abs(gr - ref_grads[k]).max() <= 1e-4 * max(1e-2, np.abs(ref_grads[k]).max()), k


def test_full_size_step_matches_chain(monkeypatch):
    """BASELINE config 2: 32 clips x 300 frames."""
    if not _supported():
        pytest.skip("needs a whole MI355X (8 XCDs x 32 CUs)")
    boxes, labels = synth.make_batch(7, 32, 300)
    m = _model()
    monkeypatch.setenv("OPNET_XCD4", "1")
    loss_x, y_x, lg_x, g_x = _run(m, boxes, labels)
    monkeypatch.setenv("OPNET_XCD4", "0")
    loss_c, y_c, lg_c, g_c = _run(m, boxes, labels)
    assert np.isfinite(y_x).all()
    assert np.abs(y_x - y_c).max() < 1e-4
    assert loss_x == pytest.approx(loss_c, abs=1e-5)
    for k in g_x:
        assert np.abs(g_x[k] - g_c[k]).max() <= 2e-3 * max(1e-2, np.abs(g_c[k]).max()), k


def test_write_through_protocol_gives_the_same_bits(monkeypatch):
    if not _supported():
        pytest.skip("needs a whole MI355X (8 XCDs x 32 CUs)")
    boxes, labels = synth.make_batch(11, 32, 20)
    m = _model()
    monkeypatch.setenv("OPNET_XCD4", "1")
    _, y_a, lg_a, g_a = _run(m, boxes, labels)
    monkeypatch.setenv("OPNET_XCD_SAFE", "1")
    _, y_b, lg_b, g_b = _run(m, boxes, labels)
    assert np.array_equal(y_a, y_b) and np.array_equal(lg_a, lg_b)
    for k in g_a:
        assert np.array_equal(g_a[k], g_b[k]), k


# ---- inference: opnet_xcd4_forward_f32 (OPNet.forward without grad, up to 32 clips) -----------------------------------------
def _infer(m, boxes, mode):
    m.use_xcd4 = mode
    with torch.no_grad():
        y, lg = m(torch.from_numpy(boxes).cuda())
    torch.cuda.synchronize()
    return y.cpu().numpy(), lg.cpu().numpy()


def test_inference_matches_reference_golden(golden_dir):
    """the reference's own OPNet.forward on 4 clips x 300 frames (tests/golden/opnet_real.npz)"""
    if not _supported():
        pytest.skip("needs a whole MI355X (8 XCDs x 32 CUs)")
    g = np.load(os.path.join(golden_dir, "opnet_real.npz"))
    boxes, _ = synth.make_batch(0, int(g["n_clips"]), int(g["t_frames"]))
    m = _model().eval()
    y, lg = _infer(m, boxes, "1")
    from objectpermanence_amd import _lib
    st = (_lib.ctypes.c_uint * 4)()
    _lib.load().opnet_xcd4_last_status(st)
    assert list(st)[:1] == [0]
    assert np.abs(y - g["y"]).max() < 1e-4
    assert np.abs(lg - g["logits"]).max() < 1e-4


@pytest.mark.parametrize("B,T", [(1, 1), (3, 5), (16, 300), (32, 64), (29, 33)])
def test_inference_matches_chain_and_oracle(B, T):
    from oracle import opnet_oracle
    if not _supported():
        pytest.skip("needs a whole MI355X (8 XCDs x 32 CUs)")
    boxes, _ = synth.make_batch(500 + B, B, T)
    m = _model().eval()
    y_x, lg_x = _infer(m, boxes, "1")
    y_x2, lg_x2 = _infer(m, boxes, "1")
    y_c, lg_c = _infer(m, boxes, "0")
    assert np.isfinite(y_x).all()
    assert np.array_equal(y_x, y_x2) and np.array_equal(lg_x, lg_x2)          # run to run: the same bits
    assert np.abs(y_x - y_c).max() < 2e-5 and np.abs(lg_x - lg_c).max() < 5e-5
    if B * T <= 2000:
        y_o, lg_o = opnet_oracle.opnet_forward(boxes, synth.opnet_synth_params(REAL_CFG), dtype=np.float64)
        assert np.abs(y_x - y_o).max() < 2e-5 and np.abs(lg_x - lg_o).max() < 5e-5


def test_inference_sees_weight_updates_and_other_streams():
    if not _supported():
        pytest.skip("needs a whole MI355X (8 XCDs x 32 CUs)")
    boxes, _ = synth.make_batch(3, 8, 20)
    m = _model().eval()
    y0, _ = _infer(m, boxes, "1")
    with torch.no_grad():
        m.prediction_layer.weight.mul_(1.5)
    y1, _ = _infer(m, boxes, "1")
    assert np.abs(y1 - 1.5 * y0).max() < 1e-5
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        y2, _ = _infer(m, boxes, "1")
    assert np.array_equal(y1, y2)


# ---- more than one row block per launch (not the default: OPNET_XCD4_MAX_B / XCD4_MAX_BATCH raise the limit) ---------------
@pytest.mark.parametrize("B,T", [(40, 9), (70, 6), (128, 4)])
def test_several_row_blocks_per_launch(B, T, monkeypatch):
    if not _supported():
        pytest.skip("needs a whole MI355X (8 XCDs x 32 CUs)")
    boxes, labels = synth.make_batch(700 + B, B, T)
    m = _model()
    monkeypatch.setenv("OPNET_XCD4", "1")
    monkeypatch.setenv("OPNET_XCD4_MAX_B", "128")
    loss_x, y_x, lg_x, g_x = _run(m, boxes, labels)
    monkeypatch.setenv("OPNET_XCD4", "0")
    loss_c, y_c, lg_c, g_c = _run(m, boxes, labels)
    assert np.isfinite(y_x).all()
    assert np.abs(y_x - y_c).max() < 2e-5 and np.abs(lg_x - lg_c).max() < 5e-5
    assert loss_x == pytest.approx(loss_c, abs=2e-6)
    for k in g_x:
        assert np.abs(g_x[k] - g_c[k]).max() <= 2e-4 * max(1e-2, np.abs(g_c[k]).max()), k
    # inference through the same kernel
    m.eval()
    m.XCD4_MAX_BATCH = 128
    y_i, lg_i = _infer(m, boxes, "1")
    y_j, lg_j = _infer(m, boxes, "0")
    assert np.abs(y_i - y_j).max() < 2e-5 and np.abs(lg_i - lg_j).max() < 5e-5
