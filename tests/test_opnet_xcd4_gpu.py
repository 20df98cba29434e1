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


def test_under_concurrent_load_and_mixed_with_the_16_clip_form():
    """A second stream keeps streaming kernels running while the persistent launches are resident; small (4-clip groups) and
    large (16-clip groups) persistent forwards and a training step are issued from different streams - the library chains
    every persistent launch of a device through one event, so no two of them are ever co-resident."""
    if not _supported():
        pytest.skip("needs a whole MI355X (8 XCDs x 32 CUs)")
    small, labels = synth.make_batch(31, 24, 40)
    large, _ = synth.make_batch(32, 200, 40)
    m = _model().eval()
    y_s, lg_s = _infer(m, small, "1")
    with torch.no_grad():
        y_l, _ = m(torch.from_numpy(large).cuda())
    torch.cuda.synchronize()
    y_l = y_l.cpu().numpy()
    m.train(True)
    _, y_t, _, g_t = _run(m, small, labels)
    xs, xl, lab = torch.from_numpy(small).cuda(), torch.from_numpy(large).cuda(), torch.from_numpy(labels).cuda()
    big = torch.empty(64 << 20, dtype=torch.float32, device="cuda:0")
    s1, s2, s3, s4 = (torch.cuda.Stream() for _ in range(4))
    torch.cuda.synchronize()
    from objectpermanence_amd import l1_mean
    outs = []
    for it in range(3):
        with torch.cuda.stream(s3):
            for _ in range(20):
                big.mul_(1.0001)
        m.eval()
        with torch.no_grad():
            with torch.cuda.stream(s1):
                outs.append(("small", m(xs)[0]))
            with torch.cuda.stream(s2):
                outs.append(("large", m(xl)[0]))
        m.train(True)
        with torch.cuda.stream(s4):
            m.zero_grad(set_to_none=True)
            yt, _ = m(xs)
            l1_mean(yt, lab).backward()
            outs.append(("train", yt.detach()))
    torch.cuda.synchronize()
    for name, y in outs:
        ref = {"small": y_s, "large": y_l, "train": y_t}[name]
        assert np.array_equal(y.cpu().numpy(), ref), name
    for k, p in m.named_parameters():
        assert np.array_equal(p.grad.cpu().numpy(), g_t[k]), k


def test_a_stuck_exchange_aborts_with_nan_and_the_next_launch_is_clean(monkeypatch):
    """OPNET_X4_DEBUG=4 switches the cells (the publishers) off: every consumer polls a sentinel that never goes away, the
    bounded wait (1.5 s) raises the abort word, every workgroup leaves, y is NaN - and nothing is left behind."""
    if not _supported():
        pytest.skip("needs a whole MI355X (8 XCDs x 32 CUs)")
    from objectpermanence_amd import _lib
    boxes, _ = synth.make_batch(41, 8, 6)
    m = _model().eval()
    y_ok, _ = _infer(m, boxes, "1")
    monkeypatch.setenv("OPNET_X4_DEBUG", "4")
    y_bad, _ = _infer(m, boxes, "1")
    st = (_lib.ctypes.c_uint * 4)()
    _lib.load().opnet_xcd4_last_status(st)
    assert st[0] == 1 and np.isnan(y_bad).all()
    monkeypatch.delenv("OPNET_X4_DEBUG")
    y_again, _ = _infer(m, boxes, "1")
    _lib.load().opnet_xcd4_last_status(st)
    assert st[0] == 0 and np.array_equal(y_again, y_ok)
