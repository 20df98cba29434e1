"""A persistent launch that gives up (bounded spins -> abort word -> NaN outputs) must be loud AND self-healing:
the drivers never write NaN-derived integers, the guarded Adam step never touches the weights, and the batch is re-run on
the launch-per-step chain (launch_monitor.py).  The aborts are forced with the tools-only debug switches of the kernels
(OPNET_XCD_DEBUG bit 3: no flag publication; OPNET_X4_DEBUG bit 2: no cells = no publishers)."""
import json
import pickle
import warnings

import numpy as np
import pytest
import torch

from oracle import opnet_oracle as oo, synth

pytestmark = pytest.mark.gpu
CFG = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}


def _model(params=None):
    from objectpermanence_amd import ModelsFactory
    params = params or synth.opnet_synth_params(CFG)
    m = ModelsFactory.get_model("opnet", CFG)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    return m.eval().to("cuda:0"), params


@pytest.mark.parametrize("B,env", [(70, "OPNET_XCD_DEBUG=8"), (12, "OPNET_X4_DEBUG=4")])
def test_aborted_forward_is_healed_in_place(monkeypatch, B, env):
    """both persistent inference forms: the launch aborts (y = NaN on the device), verify_launches() re-runs it on the chain
    into the SAME tensors, and the result equals the oracle"""
    m, params = _model()
    boxes, _ = synth.make_batch(5, B, 9)
    k, v = env.split("=")
    monkeypatch.setenv(k, v)
    with torch.no_grad():
        y, logits = m(torch.from_numpy(boxes).to("cuda:0"))
    torch.cuda.synchronize()
    assert torch.isnan(y).all()
    monkeypatch.delenv(k)
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        assert m.verify_launches() == 1
    assert m._monitor.aborted == 1 and m._monitor.healed == 1
    y_ref, lg_ref = oo.opnet_forward(boxes, params, dtype=np.float64)
    assert np.abs(y.cpu().numpy() - y_ref).max() < 2e-5 and np.abs(logits.cpu().numpy() - lg_ref).max() < 1e-4
    assert m.verify_launches() == 0                      # nothing pending any more


def test_inference_driver_writes_oracle_json_after_an_abort(tmp_path, monkeypatch):
    """reasoning_inference_main with every persistent launch forced to abort: the JSON files equal the oracle pipeline's"""
    from objectpermanence_amd.datasets import encode_boxes
    from objectpermanence_amd.inference_main import reasoning_inference_main
    s, l, out = tmp_path / "s", tmp_path / "l", tmp_path / "out"
    s.mkdir(); l.mkdir()
    raws = {}
    for i in range(6):
        name = f"v{i}"
        bb, lab, gt = synth.make_raw_video(40 + i, "plain")
        raws[name] = (bb, lab)
        pickle.dump({"bb": bb, "labels": lab}, open(s / (name + ".pkl"), "wb"), pickle.HIGHEST_PROTOCOL)
        json.dump(gt, open(l / (name + "_bb.json"), "w"))
    params = synth.opnet_synth_params(CFG)
    torch.save({k: torch.from_numpy(v) for k, v in params.items()}, tmp_path / "opnet.pth")
    json.dump(CFG, open(tmp_path / "model.json", "w"))
    json.dump({"batch_size": 2, "num_workers": 0, "device": "cuda:0", "model_path": str(tmp_path / "opnet.pth"),
               "videos_dir": "unused", "sample_dir": str(s), "labels_dir": str(l)}, open(tmp_path / "infer.json", "w"))
    monkeypatch.setenv("OPNET_X4_DEBUG", "4")            # 6 clips = one 4-clip-group persistent launch, which gives up
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        res = reasoning_inference_main("opnet", str(out), str(tmp_path / "infer.json"), str(tmp_path / "model.json"))
    boxes = np.stack([encode_boxes(*raws[n], 6).astype(np.float32) for n in res["video_names"]])
    y, _ = oo.opnet_forward(boxes, params, np.float32)
    px = oo.postprocess_to_pixels(y)
    assert (res["predictions"] != px).mean() < 2e-3 and np.abs(res["predictions"] - px).max() <= 1
    for n, p in zip(res["video_names"], res["predictions"]):
        assert np.array_equal(np.array(json.load(open(out / (n + "_bb.json")))), p)


def test_aborted_training_step_leaves_the_weights_alone_and_is_repeated(monkeypatch):
    """train_step with the forward recurrence forced to abort: gradients are NaN, the guarded Adam skips (weights and moments
    bit-identical), step_aborted() reports it and switches to the launch chain, and the repeated step equals a clean one."""
    from objectpermanence_amd import FusedAdam, _lib
    from objectpermanence_amd.training import step_aborted, train_step
    lib = _lib.load()
    boxes_np, labels_np = synth.make_batch(3, 8, 10)
    boxes, labels = torch.from_numpy(boxes_np).to("cuda:0"), torch.from_numpy(labels_np).to("cuda:0")

    def fresh():
        m, _ = _model()
        m.train(True)
        return m, FusedAdam(m.parameters(), lr=1e-3)

    try:
        # the clean step on the chain: what the repeated step must reproduce
        lib.opnet_xcd4_enable(0)
        m0, o0 = fresh()
        train_step("opnet", m0, o0, boxes, labels)
        want = [p.detach().clone() for p in m0.parameters()]
        lib.opnet_xcd4_enable(1)

        m, opt = fresh()
        before = [p.detach().clone() for p in m.parameters()]
        monkeypatch.setenv("OPNET_X4_DEBUG", "4")
        loss = train_step("opnet", m, opt, boxes, labels)
        torch.cuda.synchronize()
        monkeypatch.delenv("OPNET_X4_DEBUG")
        assert not np.isfinite(float(loss))
        for p, b in zip(m.parameters(), before):
            assert torch.equal(p.detach(), b)                       # the guard kept Adam off the weights
        with warnings.catch_warnings(record=True):
            warnings.simplefilter("always")
            assert step_aborted(m)
        assert lib.opnet_xcd4_enabled() == 0                        # this process trains on the chain from here on
        opt.rollback_step_count()
        loss2 = train_step("opnet", m, opt, boxes, labels)
        assert np.isfinite(float(loss2)) and not step_aborted(m)
        for p, w in zip(m.parameters(), want):
            assert torch.equal(p.detach(), w)
        assert all(int(st["step"]) == 1 for st in opt.state.values())
    finally:
        lib.opnet_xcd4_enable(1)


def test_non_finite_loss_skips_the_optimiser_step():
    """the guard also covers a NaN loss that no abort word announces (e.g. a NaN label)"""
    from objectpermanence_amd import FusedAdam
    from objectpermanence_amd.training import train_step
    m, _ = _model()
    m.train(True)
    opt = FusedAdam(m.parameters(), lr=1e-3)
    boxes_np, labels_np = synth.make_batch(3, 4, 6)
    labels_np = labels_np.copy()
    labels_np[0, 0, 0] = np.nan
    before = [p.detach().clone() for p in m.parameters()]
    loss = train_step("opnet", m, opt, torch.from_numpy(boxes_np).to("cuda:0"), torch.from_numpy(labels_np).to("cuda:0"))
    assert not np.isfinite(float(loss))
    for p, b in zip(m.parameters(), before):
        assert torch.equal(p.detach(), b)


def test_external_loop_with_zeroed_grads_is_not_doubled():
    """after train_step has attached a gradient bucket, an EXTERNAL loop that keeps p.grad allocated
    (zero_grad(set_to_none=False)) or accumulates two backwards must get g and g1 + g2 - not 2 g (aliased slices)."""
    from objectpermanence_amd import FusedAdam
    from objectpermanence_amd.optim import l1_mean
    from objectpermanence_amd.training import train_step
    m, _ = _model()
    m.train(True)
    opt = FusedAdam(m.parameters(), lr=0.0)
    b1, l1 = (torch.from_numpy(a).to("cuda:0") for a in synth.make_batch(1, 4, 6))
    b2, l2 = (torch.from_numpy(a).to("cuda:0") for a in synth.make_batch(9, 4, 6))
    train_step("opnet", m, opt, b1, l1)                 # attaches m._grad_bucket, p.grad aliases its slices

    def grads(batches, zero):
        zero()
        for b, l in batches:
            l1_mean(m(b)[0], l).backward()
        return [p.grad.detach().clone() for p in m.parameters()]

    g1 = grads([(b1, l1)], lambda: opt.zero_grad(set_to_none=True))
    g2 = grads([(b2, l2)], lambda: opt.zero_grad(set_to_none=True))
    kept = grads([(b1, l1)], lambda: opt.zero_grad(set_to_none=False))
    both = grads([(b1, l1), (b2, l2)], lambda: opt.zero_grad(set_to_none=True))
    for a, b, k, s in zip(g1, g2, kept, both):
        assert torch.allclose(k, a, rtol=0, atol=1e-7 * float(a.abs().max()) + 1e-12)
        assert torch.allclose(s, a + b, rtol=0, atol=2e-6 * float((a + b).abs().max()) + 1e-12)
