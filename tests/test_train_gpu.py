"""GPU parity of the training path (BPTT kernels, weight-gradient GEMMs, L1 loss, Adam) against
gradients / Adam steps produced by the reference's own model under torch autograd
(tests/golden/opnet_train_*.npz) and against oracle/torch_port.py at other shapes."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import synth, torch_port

pytestmark = pytest.mark.gpu

REAL_CFG = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}


def sample_indices(name, n, k=4096):
    if n <= k:
        return np.arange(n)
    u = synth.counter_uniform(synth.name_seed(name, 99), k)
    return np.unique((u * n).astype(np.int64))


def _model(cfg):
    from objectpermanence_amd import ModelsFactory
    m = ModelsFactory.get_model("opnet", cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.opnet_synth_params(cfg).items()})
    return m.to("cuda:0").train(True)


def _step_grads(m, boxes, labels):
    from objectpermanence_amd import l1_mean
    m.zero_grad(set_to_none=True)
    y, logits = m(torch.from_numpy(boxes).cuda())
    loss = l1_mean(y, torch.from_numpy(labels).cuda())
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.item()), {k: p.grad.cpu().numpy() for k, p in m.named_parameters()}, y.detach().cpu().numpy()


def test_tiny_training_matches_reference(golden_dir):
    from objectpermanence_amd import FusedAdam
    g = np.load(os.path.join(golden_dir, "opnet_train_tiny.npz"))
    cfg = json.loads(str(g["cfg"]))
    boxes, labels = synth.make_batch(0, int(g["n_clips"]), int(g["t_frames"]))
    m = _model(cfg)
    opt = FusedAdam(m.parameters(), lr=1e-3)
    for step, ref_loss in enumerate(g["losses"]):
        loss, grads, _ = _step_grads(m, boxes, labels)
        assert loss == pytest.approx(float(ref_loss), abs=5e-6)
        if step == 0:
            for k, gr in grads.items():
                ref = g["grad/" + k]
                assert np.abs(gr - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), k
        opt.step()
    torch.cuda.synchronize()
    for k, p in m.named_parameters():
        assert np.abs(p.detach().cpu().numpy() - g["w_after/" + k]).max() < 5e-5, k


def test_real_gradients_match_reference(golden_dir):
    from objectpermanence_amd import FusedAdam
    g = np.load(os.path.join(golden_dir, "opnet_train_real.npz"))
    boxes, labels = synth.make_batch(0, int(g["n_clips"]), int(g["t_frames"]))
    m = _model(REAL_CFG)
    opt = FusedAdam(m.parameters(), lr=1e-3)
    for step, ref_loss in enumerate(g["losses"]):
        loss, grads, _ = _step_grads(m, boxes, labels)
        assert loss == pytest.approx(float(ref_loss), abs=2e-5 if step == 0 else 2e-3)
        if step == 0:
            for k, gr in grads.items():
                idx, ref = g["gidx/" + k], g["gval/" + k]
                scale = max(1e-3, np.abs(ref).max())
                assert np.abs(gr.reshape(-1)[idx] - ref).max() <= 5e-4 * scale, k
                assert np.sqrt((gr.astype(np.float64) ** 2).sum()) == pytest.approx(float(g["gnorm/" + k]), rel=2e-4)
        opt.step()
    torch.cuda.synchronize()
    for k, p in m.named_parameters():
        w = p.detach().cpu().numpy().reshape(-1)[sample_indices(k, p.numel())]
        # two Adam steps move every weight by <= 2e-3; a sign flip of a ~zero gradient moves it by 1e-3
        assert np.abs(w - g["w_after_val/" + k]).max() < 2.1e-3, k
        assert np.mean(np.abs(w - g["w_after_val/" + k]) < 2e-5) > 0.98, k


@pytest.mark.parametrize("B,T", [(1, 1), (2, 5), (33, 6), (70, 3), (131, 4)])   # 131: wide step kernel + split backward
def test_gradients_match_torch_port_ragged(B, T):
    boxes, labels = synth.make_batch(200, B, T)
    m = _model(REAL_CFG)
    loss, grads, y = _step_grads(m, boxes, labels)
    ref_loss, ref_grads, y_ref = torch_port.loss_and_grads(boxes, labels, synth.opnet_synth_params(REAL_CFG),
                                                           dtype=torch.float64)
    assert np.abs(y - y_ref).max() < 2e-5
    assert loss == pytest.approx(ref_loss, abs=2e-6)
    for k, gr in grads.items():
        ref = ref_grads[k]
        assert np.abs(gr - ref).max() <= 1e-4 * max(1e-2, np.abs(ref).max()), k


def test_train_forward_equals_inference_forward():
    boxes, _ = synth.make_batch(5, 6, 50)
    m = _model(REAL_CFG)
    x = torch.from_numpy(boxes).cuda()
    y_t, l_t = m(x)
    with torch.no_grad():
        y_i, l_i = m(x)
    torch.cuda.synchronize()
    # 6 clips train on the 4-clip persistent step (another summation order than the inference chain's)
    assert (y_t.detach() - y_i).abs().max() < 2e-5 and (l_t - l_i).abs().max() < 5e-5
    assert y_t.requires_grad and not l_t.requires_grad


def test_train_forward_equals_inference_forward_on_the_chain(monkeypatch):
    monkeypatch.setenv("OPNET_XCD4", "0")
    boxes, _ = synth.make_batch(5, 6, 50)
    m = _model(REAL_CFG)
    x = torch.from_numpy(boxes).cuda()
    y_t, l_t = m(x)
    with torch.no_grad():
        y_i, l_i = m(x)
    torch.cuda.synchronize()
    assert torch.equal(y_t.detach(), y_i) and torch.equal(l_t, l_i)


def test_backward_after_second_forward_is_refused():
    from objectpermanence_amd import l1_mean
    boxes, labels = synth.make_batch(5, 2, 8)
    m = _model(REAL_CFG)
    x, lab = torch.from_numpy(boxes).cuda(), torch.from_numpy(labels).cuda()
    y1, _ = m(x)
    y2, _ = m(x)
    with pytest.raises(RuntimeError, match="overwritten"):
        l1_mean(y1, lab).backward()
    l1_mean(y2, lab).backward()


def test_l1_mean_and_adam_match_torch():
    from objectpermanence_amd import FusedAdam, l1_mean
    torch.manual_seed(0)
    y = torch.randn(7, 13, 4, device="cuda", requires_grad=True)
    lab = torch.randn(7, 13, 4, device="cuda")
    lab[0, 0, 0] = y.detach()[0, 0, 0]          # exercise sign(0) = 0
    loss = l1_mean(y, lab)
    loss.backward()
    y2 = y.detach().clone().requires_grad_(True)
    ref = torch.mean(torch.nn.L1Loss(reduction="none")(y2, lab))
    ref.backward()
    assert float(loss) == pytest.approx(float(ref), rel=1e-6)
    assert torch.equal(y.grad, y2.grad)
    p1 = torch.nn.Parameter(torch.randn(1000, device="cuda"))
    p2 = torch.nn.Parameter(p1.detach().clone())
    o1, o2 = FusedAdam([p1], lr=1e-3), torch.optim.Adam([p2], lr=1e-3)
    for _ in range(5):
        gr = torch.randn(1000, device="cuda")
        p1.grad, p2.grad = gr.clone(), gr.clone()
        o1.step(); o2.step()
    assert torch.allclose(p1, p2, rtol=0, atol=2e-7)


@pytest.mark.parametrize("xcd4", ["0", "1"])
def test_inference_after_fused_adam_uses_updated_weights(xcd4, monkeypatch):
    """Regression: FusedAdam updates parameters outside torch's view; the inference path's packed-weight
    cache is keyed on the parameter version and must notice."""
    from objectpermanence_amd import FusedAdam, l1_mean
    monkeypatch.setenv("OPNET_XCD4", xcd4)
    boxes, labels = synth.make_batch(9, 4, 20)
    m = _model(REAL_CFG)
    x, lab = torch.from_numpy(boxes).cuda(), torch.from_numpy(labels).cuda()
    opt = FusedAdam(m.parameters(), lr=1e-2)
    with torch.no_grad():
        y0, _ = m(x)
    for _ in range(3):
        opt.zero_grad(set_to_none=True)
        y, _ = m(x)
        l1_mean(y, lab).backward()
        opt.step()
    with torch.no_grad():
        y1, _ = m(x)                      # inference path after training steps
    yt, _ = m(x)                          # training path (always repacks)
    torch.cuda.synchronize()
    assert not torch.equal(y0, y1)
    if xcd4 == "0":
        assert torch.equal(y1, yt.detach())
    else:       # 4 clips train on the 4-clip persistent step: another summation order than the inference chain's
        assert (y1 - yt.detach()).abs().max() < 2e-5 and (y0 - y1).abs().max() > 1e-3


def test_smooth_l1_matches_torch():
    from objectpermanence_amd.optim import smooth_l1_mean
    torch.manual_seed(1)
    y = (torch.randn(5, 11, 4, device="cuda") * 0.7).requires_grad_(True)
    lab = torch.randn(5, 11, 4, device="cuda") * 0.7
    for beta in (1.0, 0.25):
        y.grad = None
        loss = smooth_l1_mean(y, lab, beta)
        loss.backward()
        y2 = y.detach().clone().requires_grad_(True)
        ref = torch.nn.SmoothL1Loss(beta=beta)(y2, lab)
        ref.backward()
        assert float(loss.detach()) == pytest.approx(float(ref.detach()), rel=2e-6)
        assert torch.allclose(y.grad, y2.grad, rtol=1e-6, atol=1e-9)


def test_fused_and_split_backward_agree(monkeypatch):
    """the two reverse-recurrence schedules (one fused launch per step; the split-K pair of launches, which the library now only
    picks beyond 16 row blocks a slice) give the same gradients; the automatic choice - from two row blocks on, slices of the
    batch as separate chains of the fused step on separate streams - is the fused schedule bit for bit"""
    from objectpermanence_amd import ModelsFactory, l1_mean
    cfg = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 48, "videos_hidden_dim": 64}
    p = synth.opnet_synth_params(cfg)

    def grads(B, mode):
        if mode:
            monkeypatch.setenv("OPNET_BWD_MODE", mode)
        else:
            monkeypatch.delenv("OPNET_BWD_MODE", raising=False)
        boxes, labels = synth.make_batch(31, B, 6)
        m = ModelsFactory.get_model("opnet", cfg)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in p.items()})
        m.to("cuda:0").train(True)
        l1_mean(m(torch.from_numpy(boxes).cuda())[0], torch.from_numpy(labels).cuda()).backward()
        torch.cuda.synchronize()
        return {k: v.grad.cpu().numpy() for k, v in m.named_parameters()}

    for B in (5, 70, 150):
        gf, gs, ga = grads(B, "fused"), grads(B, "split"), grads(B, None)
        for k in gf:
            scale = max(1e-3, np.abs(gs[k]).max())
            assert np.abs(gf[k] - gs[k]).max() <= 2e-5 * scale, (B, k)
            assert np.array_equal(ga[k], gf[k]), (B, k)       # the automatic choice


@pytest.mark.parametrize("B,T", [(32, 40), (7, 13), (70, 9)])
def test_wave_tile_weight_gradients_agree_with_the_workgroup_tile_kernel(monkeypatch, B, T):
    """opnet_wgrad_tiles + opnet_wgrad_reduce (one wave per 128 x 128 / 64 x 64 tile and time slice) against
    opnet_wgrad (OPNET_WGRAD2=0) on the same histories: the reference hidden sizes (both tile forms, ragged row blocks), same
    gradients up to the order of the sum over time; and the wave-tile form reproduces itself bit for bit"""
    from objectpermanence_amd import ModelsFactory, l1_mean
    cfg = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}
    p = synth.opnet_synth_params(cfg)
    boxes, labels = synth.make_batch(5, B, T)

    def grads(flag):
        monkeypatch.setenv("OPNET_WGRAD2", flag)
        m = ModelsFactory.get_model("opnet", cfg)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in p.items()})
        m.to("cuda:0").train(True)
        l1_mean(m(torch.from_numpy(boxes).cuda())[0], torch.from_numpy(labels).cuda()).backward()
        torch.cuda.synchronize()
        return {k: v.grad.cpu().numpy() for k, v in m.named_parameters()}

    new, old, again = grads("1"), grads("0"), grads("1")
    for k in new:
        scale = max(1e-6, np.abs(old[k]).max())
        assert np.isfinite(new[k]).all() and np.abs(new[k] - old[k]).max() <= 2e-5 * scale, k
        assert np.array_equal(new[k], again[k]), k
