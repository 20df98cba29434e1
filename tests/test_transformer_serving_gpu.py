"""Independent `transformer_lstm` requests served in ONE pass (config 3 throughput).

The reference runs one forward per request (baselines/learned_models.py:176-197): attention spans the S = b * T tokens of THAT
call.  `TransformerLstm.forward_segments` / `ReasonerServer` merge pending requests - token-wise stages over all tokens,
attention inside a request (opseq_encoder_layer_segmented_f32), one persistent stacked-LSTM launch over all clips - and must
hand every request the bits its lone forward produces; and those bits are pinned to the reference by the goldens."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import synth

pytestmark = pytest.mark.gpu
REAL = {"boxes_features_dim": 256, "num_attention_heads": 2, "num_attention_layers": 2, "num_lstm_layers": 2, "lstm_hidden_dim": 512}


def _model(cfg, engine="auto"):
    from objectpermanence_amd import ModelsFactory
    m = ModelsFactory.get_model("transformer_lstm", cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.transformer_lstm_synth_params(cfg).items()})
    if engine == "chain":
        m._runner.use_xcd = "0"
    return m.eval().to("cuda:0")


def _requests(first, n, b, T):
    return [torch.from_numpy(synth.boxes5(synth.make_batch(first + 7 * r, b, T)[0])).cuda() for r in range(n)]


# (4, 8, 4, 300): eight coupled requests of four clips (S = 1 200 each)
@pytest.mark.parametrize("heads,n,b,T", [(2, 16, 1, 300), (4, 16, 1, 300), (4, 5, 2, 300), (2, 7, 1, 37), (4, 3, 3, 50), (2, 32, 1, 20),
                                         (4, 8, 4, 300)])
def test_every_request_of_a_merged_pass_is_bit_identical_to_its_lone_forward(heads, n, b, T):
    cfg = dict(REAL, num_attention_heads=heads)
    m = _model(cfg)
    reqs = _requests(100, n, b, T)
    with torch.no_grad():
        alone = [m(r).clone() for r in reqs]
        merged = m.forward_segments(torch.cat(reqs), n)
    torch.cuda.synchronize()
    assert m._runner._monitor.verify() == 0
    assert merged.shape == (n * b, T, 4)
    for r in range(n):
        assert torch.equal(merged[r * b:(r + 1) * b], alone[r]), r
    if n > 1 and b * T > 16:
        # ... and it IS segmented: the same clips as ONE coupled minibatch give something else (reference quirk: S = B * T)
        with torch.no_grad():
            coupled = m(torch.cat(reqs))
        assert not torch.equal(coupled[:b], alone[0])


@pytest.mark.parametrize("tag", ["real_b1", "heads4_b1", "real_b2"])
def test_merged_requests_match_the_reference_golden(golden_dir, tag):
    """the golden request (outputs of the reference's own class) rides in a pass with 8 other requests and still matches"""
    gold = np.load(os.path.join(golden_dir, "siblings.npz"))
    cfg = json.loads(str(gold[f"transformer_lstm/{tag}/cfg"]))
    n, t = (int(v) for v in gold[f"transformer_lstm/{tag}/shape"])
    x = torch.from_numpy(synth.boxes5(synth.make_batch(0, n, t)[0])).cuda()
    others = _requests(500, 8, n, t)
    m = _model(cfg)
    with torch.no_grad():
        y = m.forward_segments(torch.cat(others[:3] + [x] + others[3:]), 9)[3 * n:4 * n]
    torch.cuda.synchronize()
    y_ref = gold[f"transformer_lstm/{tag}/y"]
    assert np.abs(y.cpu().numpy() - y_ref).max() < 3e-5


def test_server_merges_transformer_requests_as_segments():
    from objectpermanence_amd.serving import ReasonerServer
    m = _model(dict(REAL, num_attention_heads=4))
    reqs = _requests(900, 6, 1, 300) + _requests(950, 2, 2, 300)      # six one-clip requests, then two of another shape
    with torch.no_grad():
        alone = [m(r).clone() for r in reqs]
    launches0 = m._runner.xcd_launches
    server = ReasonerServer(m, "transformer_lstm")
    handles = [server.submit(r) for r in reqs]
    outs = [h.result() for h in handles]
    torch.cuda.synchronize()
    assert server.forwards == 2 and m._runner.xcd_launches - launches0 == 2       # one pass per request shape
    for o, a in zip(outs, alone):
        assert torch.equal(o, a)


def test_two_passes_in_flight_return_the_lone_results():
    """a segmented server issues its passes on two side streams in turn (one pass's encoder under the other's recurrence): 80
    one-clip requests = five passes of 16; every result - read in any order, consumed on the caller's stream with no host wait -
    is the lone forward's, and a one-stream server gives the same"""
    from objectpermanence_amd.serving import ReasonerServer
    m = _model(dict(REAL, num_attention_heads=4))
    reqs = _requests(2100, 80, 1, 300)
    with torch.no_grad():
        alone = [m(r).clone() for r in reqs]
    torch.cuda.synchronize()
    for streams in (None, 1):
        server = ReasonerServer(m, "transformer_lstm", max_clips=16, streams=streams)
        handles = [server.submit(r) for r in reqs]
        assert server.forwards == 5 and all(h.done() for h in handles)
        assert (handles[0]._event is not None) == (streams is None)
        total = torch.zeros((), device="cuda:0")
        for k in list(range(79, -1, -7)) + list(range(80)):                      # any order, some twice
            y = handles[k].result()
            assert handles[k]._event is None
            total = total + (y - alone[k]).abs().sum()                              # enqueued on the caller's stream, no sync before
        torch.cuda.synchronize()
        assert float(total) == 0.0
        for h, a in zip(handles, alone):
            assert torch.equal(h.result(), a)


def test_a_pass_is_cut_where_the_lone_engine_would_change():
    """the persistent stack launch carries 128 clips (L = 2): the server flushes there instead of letting the merged batch fall
    to the launch chain, whose sums differ in the last bits from the lone request's"""
    from objectpermanence_amd import _lib
    from objectpermanence_amd.serving import ReasonerServer
    m = _model(REAL)
    cap = int(_lib.load().opseq_xcd_max_batch(2))
    assert m.max_requests_per_pass(1, 10, exact=True) == cap and m.max_requests_per_pass(3, 10, exact=True) == cap // 3
    reqs = _requests(1200, cap + 3, 1, 10)
    with torch.no_grad():
        alone = [m(r).clone() for r in reqs[:2] + reqs[-2:]]
    server = ReasonerServer(m, "transformer_lstm", max_clips=4096, exact=True)
    handles = [server.submit(r) for r in reqs]
    server.flush()
    assert server.forwards == 2
    for h, a in zip(handles[:2] + handles[-2:], alone):
        assert torch.equal(h.result(), a)
    with pytest.raises(ValueError, match="exceed one pass"), torch.no_grad():
        m.forward_segments(torch.cat(reqs), len(reqs), exact=True)


@pytest.mark.parametrize("heads,n,b,T", [(2, 70, 1, 40), (4, 40, 2, 30), (2, 130, 1, 12)])
def test_a_large_pass_takes_the_throughput_form_and_agrees_to_rounding(heads, n, b, T):
    """_LstmStackRunner.XCDT_MIN_BATCH clips or more in one pass (exact = False, the default): token-wise products on large tiles, the stacked LSTM on 16-clip
    groups (csrc/seq_xcdt_kernels.hip) - every request agrees with its lone forward to rounding, the pass reproduces itself bit
    for bit, exact = True still returns the lone bits (cut into passes the 4-clip launch carries), and requests do not mix"""
    cfg = dict(REAL, num_attention_heads=heads)
    m = _model(cfg)
    reqs = _requests(300, n, b, T)
    lib_cap = m.max_requests_per_pass(b, T)
    assert lib_cap >= n
    with torch.no_grad():
        alone = [m(r).clone() for r in reqs]
        before = m._runner.xcdt_launches
        merged = m.forward_segments(torch.cat(reqs), n)
        assert m.last_pass_engine == "t" and m._runner.xcdt_launches == before + 1
        again = m.forward_segments(torch.cat(reqs), n)
        # the same requests in another order: each still gets its own result (segments do not see each other)
        perm = list(range(n))[::-1]
        shuffled = m.forward_segments(torch.cat([reqs[i] for i in perm]), n)
    torch.cuda.synchronize()
    assert m._runner._monitor.verify() == 0
    assert torch.equal(merged, again)
    worst = max(float((merged[r * b:(r + 1) * b] - alone[r]).abs().max()) for r in range(n))
    assert worst < 1e-5, worst
    for k, i in enumerate(perm):
        assert float((shuffled[k * b:(k + 1) * b] - merged[i * b:(i + 1) * b]).abs().max()) < 1e-5
    cap = m.max_requests_per_pass(b, T, exact=True)
    with torch.no_grad():
        for lo in range(0, n, cap):
            part = m.forward_segments(torch.cat(reqs[lo:lo + cap]), len(reqs[lo:lo + cap]), exact=True) if len(reqs[lo:lo + cap]) > 1 else m(reqs[lo])
            for k, r in enumerate(range(lo, min(n, lo + cap))):
                assert torch.equal(part[k * b:(k + 1) * b], alone[r])


@pytest.mark.parametrize("tag", ["real_b1", "heads4_b1", "real_b2"])
def test_throughput_pass_matches_the_reference_golden(golden_dir, tag):
    """the golden request (outputs of the reference's own class) rides in a throughput pass of 70 requests and still matches"""
    gold = np.load(os.path.join(golden_dir, "siblings.npz"))
    cfg = json.loads(str(gold[f"transformer_lstm/{tag}/cfg"]))
    n, t = (int(v) for v in gold[f"transformer_lstm/{tag}/shape"])
    x = torch.from_numpy(synth.boxes5(synth.make_batch(0, n, t)[0])).cuda()
    others = _requests(500, 69, n, t)
    m = _model(cfg)
    with torch.no_grad():
        y = m.forward_segments(torch.cat(others[:30] + [x] + others[30:]), 70)[30 * n:31 * n]
    torch.cuda.synchronize()
    assert m.last_pass_engine == "t"
    y_ref = gold[f"transformer_lstm/{tag}/y"]
    assert np.abs(y.cpu().numpy() - y_ref).max() < 3e-5


def test_inference_driver_serves_transformer_minibatches_as_segments(tmp_path):
    """reasoning_inference_main on transformer_lstm: every DataLoader minibatch is one request (the reference's call), merged
    by the server - the predictions equal a loop of plain per-minibatch forwards"""
    import pickle
    from objectpermanence_amd import metrics
    from objectpermanence_amd.datasets import DatasetsFactory
    from objectpermanence_amd.inference_main import reasoning_inference_main
    s, l = tmp_path / "s", tmp_path / "l"
    s.mkdir(); l.mkdir()
    for i in range(7):
        bb, lab, gt = synth.make_raw_video(60 + i, "plain")
        pickle.dump({"bb": bb, "labels": lab}, open(s / f"v{i}.pkl", "wb"), pickle.HIGHEST_PROTOCOL)
        json.dump(gt, open(l / f"v{i}_bb.json", "w"))
    cfg = dict(REAL, num_attention_heads=4)
    params = synth.transformer_lstm_synth_params(cfg)
    torch.save({k: torch.from_numpy(v) for k, v in params.items()}, tmp_path / "t.pth")
    json.dump(cfg, open(tmp_path / "model.json", "w"))
    json.dump({"batch_size": 2, "num_workers": 0, "device": "cuda:0", "model_path": str(tmp_path / "t.pth"),
               "videos_dir": "unused", "sample_dir": str(s), "labels_dir": str(l)}, open(tmp_path / "infer.json", "w"))
    res = reasoning_inference_main("transformer_lstm", str(tmp_path / "out"), str(tmp_path / "infer.json"), str(tmp_path / "model.json"))
    ds = DatasetsFactory.get_inference_dataset("transformer_lstm", str(s), str(l))
    m = _model(cfg)
    want = []
    for lo in range(0, 7, 2):
        x = torch.stack([ds[i][0][0] for i in range(lo, min(7, lo + 2))]).cuda()
        with torch.no_grad():
            want.append(metrics.postprocess_and_iou(m(x))[0].cpu().numpy())
    assert np.array_equal(res["predictions"], np.concatenate(want))
