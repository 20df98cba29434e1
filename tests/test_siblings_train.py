"""Training parity of the sibling reasoners (BaselineLstm, NonLinearLstm, OPNetLstmMlp): gradients of the L1 loss against
the reference's own models under torch autograd (tests/golden/siblings_train.npz); oracle/torch_port.py pinned
against the same fixtures on CPU."""
import json
import os

import numpy as np
import pytest

from oracle import synth, torch_port

PARAMS = {"baseline_lstm": synth.baseline_lstm_synth_params, "non_linear_lstm": synth.non_linear_lstm_synth_params,
          "opnet_lstm_mlp": synth.opnet_lstm_mlp_synth_params, "transformer_lstm": synth.transformer_lstm_synth_params}
# transformer_lstm: the reference in train mode with its dropout probabilities set to 0 (its torch masks cannot be
# reproduced); the HIP model is run with dropout = 0.0 accordingly
CASES = [("baseline_lstm", "tiny"), ("baseline_lstm", "real"), ("non_linear_lstm", "tiny"), ("non_linear_lstm", "real"),
         ("opnet_lstm_mlp", "tiny"), ("opnet_lstm_mlp", "real"), ("transformer_lstm", "tiny"), ("transformer_lstm", "real")]


def _features(name, boxes):
    return boxes if name == "opnet_lstm_mlp" else synth.boxes5(boxes)


def _y(out):
    return out[0] if isinstance(out, tuple) else out


def sample_indices(name, n, k=4096):
    if n <= k:
        return np.arange(n)
    u = synth.counter_uniform(synth.name_seed(name, 99), k)
    return np.unique((u * n).astype(np.int64))


def _case(g, name, tag):
    pre = f"{name}/{tag}/"
    cfg = json.loads(str(g[pre + "cfg"]))
    n, t = (int(v) for v in g[pre + "shape"])
    boxes, labels = synth.make_batch(0, n, t)
    return pre, cfg, _features(name, boxes), labels


def _check(g, pre, loss, grads, rel):
    assert loss == pytest.approx(float(g[pre + "loss"]), abs=5e-6)
    for k, gr in grads.items():
        ref = g[pre + "gval/" + k]
        got = gr.reshape(-1)[sample_indices(k, gr.size)]
        scale = max(1e-3, np.abs(ref).max())
        assert np.abs(got - ref).max() <= rel * scale, k
        assert np.sqrt((gr.astype(np.float64) ** 2).sum()) == pytest.approx(float(g[pre + "gnorm/" + k]), rel=5e-4), k


@pytest.mark.parametrize("name,tag", CASES)
def test_torch_port_matches_reference(golden_dir, name, tag):
    g = np.load(os.path.join(golden_dir, "siblings_train.npz"))
    pre, cfg, x, labels = _case(g, name, tag)
    loss, grads, _ = torch_port.sibling_loss_and_grads(name, x, labels, PARAMS[name](cfg),
                                                       nhead=cfg.get("num_attention_heads", 2))
    _check(g, pre, loss, grads, 2e-4)


@pytest.fixture
def stack_engine(request):
    """"auto": the product's choice - the H = 512 stacks run their forward (inference AND training) as one persistent launch
    (csrc/seq_xcd_kernels.hip); "chain": one launch per time step (csrc/seq_kernels.hip) for everything"""
    from objectpermanence_amd import _lib
    lib = _lib.load()
    lib.opseq_xcd_enable(0 if request.param == "chain" else 1)
    yield request.param
    lib.opseq_xcd_enable(1)


@pytest.mark.gpu
@pytest.mark.parametrize("stack_engine", ["auto", "chain"], indirect=True)
@pytest.mark.parametrize("name,tag", CASES)
def test_hip_gradients_match_reference(golden_dir, name, tag, stack_engine):
    import torch
    from objectpermanence_amd import ModelsFactory, l1_mean
    g = np.load(os.path.join(golden_dir, "siblings_train.npz"))
    pre, cfg, x, labels = _case(g, name, tag)
    m = ModelsFactory.get_model(name, cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in PARAMS[name](cfg).items()})
    m.to("cuda:0").train(True)
    if name == "transformer_lstm":
        m.dropout = 0.0
    y = _y(m(torch.from_numpy(x).cuda()))
    loss = l1_mean(y, torch.from_numpy(labels).cuda())
    loss.backward()
    torch.cuda.synchronize()
    _check(g, pre, float(loss.detach()), {k: p.grad.cpu().numpy() for k, p in m.named_parameters()}, 5e-4)
    if hasattr(m, "_runner"):
        persistent = stack_engine == "auto" and tag == "real"
        # the training forward AND the reverse recurrence were watched <=> persistent (an entry whose launch has completed clean
        # may already have been dropped by the second watch)
        assert m._runner._monitor.pending() in ((1, 2) if persistent else (0,))
        assert not m.training_step_aborted()
    with torch.no_grad():
        y_inf = _y(m(torch.from_numpy(x).cuda()))
    if name == "transformer_lstm":      # materialised-softmax training kernels vs the flash inference kernel
        assert (y_inf - y.detach()).abs().max().item() < 1e-4
    else:
        assert torch.equal(y_inf, y.detach())          # train-mode forward == inference forward, bit for bit


@pytest.mark.gpu
@pytest.mark.parametrize("name,B,T", [("baseline_lstm", 33, 5), ("non_linear_lstm", 2, 3), ("opnet_lstm_mlp", 37, 4)])
def test_hip_gradients_ragged_vs_torch_port(name, B, T):
    import torch
    from objectpermanence_amd import ModelsFactory, l1_mean
    cfg = {"baseline_lstm": {"videos_hidden_dim": 64}, "non_linear_lstm": {"boxes_features_dim": 16, "videos_hidden_dim": 48},
           "opnet_lstm_mlp": {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 48, "videos_hidden_dim": 64}}[name]
    boxes, labels = synth.make_batch(77, B, T)
    x = _features(name, boxes)
    p = PARAMS[name](cfg)
    m = ModelsFactory.get_model(name, cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in p.items()})
    m.to("cuda:0").train(True)
    loss = l1_mean(_y(m(torch.from_numpy(x).cuda())), torch.from_numpy(labels).cuda())
    loss.backward()
    ref_loss, ref, _ = torch_port.sibling_loss_and_grads(name, x, labels, p, dtype=torch.float64)
    assert float(loss.detach()) == pytest.approx(ref_loss, abs=2e-6)
    for k, prm in m.named_parameters():
        assert np.abs(prm.grad.cpu().numpy() - ref[k]).max() <= 1e-4 * max(1e-2, np.abs(ref[k]).max()), k


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,E,nhead", [(3, 7, 64, 4), (1, 33, 32, 2)])
def test_transformer_gradients_ragged_vs_torch_port(B, T, E, nhead):
    """S = B*T not a multiple of 16, 4 heads / head size 16: encoder + LSTM + embedding gradients vs fp64 autograd"""
    import torch
    from objectpermanence_amd import ModelsFactory, l1_mean
    cfg = {"boxes_features_dim": E, "num_attention_heads": nhead, "num_attention_layers": 2, "num_lstm_layers": 2,
           "lstm_hidden_dim": 48}
    boxes, labels = synth.make_batch(55, B, T)
    x = synth.boxes5(boxes)
    p = synth.transformer_lstm_synth_params(cfg)
    m = ModelsFactory.get_model("transformer_lstm", cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in p.items()})
    m.to("cuda:0").train(True)
    m.dropout = 0.0
    loss = l1_mean(m(torch.from_numpy(x).cuda()), torch.from_numpy(labels).cuda())
    loss.backward()
    ref_loss, ref, _ = torch_port.sibling_loss_and_grads("transformer_lstm", x, labels, p, dtype=torch.float64, nhead=nhead)
    assert float(loss.detach()) == pytest.approx(ref_loss, abs=3e-6)
    for k, prm in m.named_parameters():
        assert np.abs(prm.grad.cpu().numpy() - ref[k]).max() <= 2e-4 * max(1e-2, np.abs(ref[k]).max()), k


@pytest.mark.gpu
def test_transformer_dropout_masks_and_backward_consistency():
    """train mode with the reference's dropout 0.1: masks come from a counter generator (not torch's, so no parity):
    same seed -> same output, another call -> another mask, about 10 % of the FFN activations dropped in expectation,
    and the backward uses the SAME masks as the forward (directional derivative of the fixed-mask loss)"""
    import torch
    from objectpermanence_amd import ModelsFactory, l1_mean
    cfg = {"boxes_features_dim": 32, "num_attention_heads": 2, "num_attention_layers": 2, "num_lstm_layers": 2,
           "lstm_hidden_dim": 32}
    boxes, labels = synth.make_batch(5, 2, 12)
    x, lab = torch.from_numpy(synth.boxes5(boxes)).cuda(), torch.from_numpy(labels).cuda()
    m = ModelsFactory.get_model("transformer_lstm", cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.transformer_lstm_synth_params(cfg).items()})
    m.to("cuda:0").train(True)
    assert m.dropout == pytest.approx(0.1)

    def loss_at(calls):
        m._calls = calls                       # the mask stream is keyed by (dropout_seed, call counter, layer)
        return l1_mean(m(x), lab)

    l0, l0b, l1 = loss_at(7), loss_at(7), loss_at(8)
    assert float(l0.detach()) == float(l0b.detach()) and float(l0.detach()) != float(l1.detach())
    m.eval()
    with torch.no_grad():
        y_eval = m(x)
    m.train(True)
    m._calls = 7
    assert (m(x).detach() - y_eval).abs().max().item() > 1e-4          # dropout does change the output
    # directional derivative with the masks frozen (same call counter)
    m.zero_grad()
    loss_at(7).backward()
    prm = m.attention_encoder.layers[0].linear1.weight
    g = prm.grad.detach().clone()
    d = torch.randn_like(prm)
    d /= d.norm()
    eps = 2e-2
    def shifted(step):
        with torch.no_grad():
            prm.add_(step * d)
        try:
            return float(loss_at(7).detach())
        finally:
            with torch.no_grad():
                prm.sub_(step * d)

    lp, lm_ = shifted(eps), shifted(-eps)
    num, ana = (lp - lm_) / (2 * eps), float((g * d).sum())
    assert abs(num - ana) <= 0.15 * max(abs(ana), 1e-3) + 2e-4, (num, ana)
    with pytest.raises(RuntimeError):
        with torch.no_grad():
            m(x)                               # train mode without gradients would silently apply dropout: refuse


@pytest.mark.gpu
@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_chunked_attention_training_is_chunk_size_independent(monkeypatch, p_drop):
    """The training encoder evaluates attention in chunks of query rows and recomputes the chunk's probabilities in the
    backward (no S x S matrix is kept).  With the chunk forced to 16 / 48 rows on S = 3 x 37 = 111 tokens (ragged last chunk)
    loss and every gradient must equal the single-chunk run up to the summation order of dK / dV over chunks - with and
    without dropout (the masks are keyed by the element's global index, not by the chunk)."""
    import torch
    from objectpermanence_amd import ModelsFactory, l1_mean
    cfg = {"boxes_features_dim": 64, "num_attention_heads": 4, "num_attention_layers": 2, "num_lstm_layers": 2,
           "lstm_hidden_dim": 48}
    boxes, labels = synth.make_batch(21, 3, 37)
    x, lab = torch.from_numpy(synth.boxes5(boxes)).cuda(), torch.from_numpy(labels).cuda()

    monkeypatch.setenv("OPSEQ_ATTN_FLASH", "0")        # the chunked GEMM form (what head sizes outside 16 / 32 / 64 / 128 run)

    def run(chunk):
        if chunk:
            monkeypatch.setenv("OPSEQ_ATTN_CHUNK", str(chunk))
        else:
            monkeypatch.delenv("OPSEQ_ATTN_CHUNK", raising=False)
        m = ModelsFactory.get_model("transformer_lstm", cfg)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.transformer_lstm_synth_params(cfg).items()})
        m.to("cuda:0").train(True)
        m.dropout = p_drop
        m._calls = 3
        loss = l1_mean(m(x), lab)
        loss.backward()
        torch.cuda.synchronize()
        return float(loss.detach()), {k: p.grad.cpu().numpy() for k, p in m.named_parameters()}

    l_ref, g_ref = run(0)
    for chunk in (16, 48):
        l, g = run(chunk)
        assert l == pytest.approx(l_ref, abs=1e-6)
        for k in g_ref:
            assert np.abs(g[k] - g_ref[k]).max() <= 2e-5 * max(1e-2, np.abs(g_ref[k]).max()), (chunk, k)


@pytest.mark.gpu
def test_long_sequence_training_step_runs_in_bounded_memory():
    """S = 38 400 tokens (128 clips x 300 frames, 2 heads): round 2's materialised S x S softmax stopped at S ~ 23 000 and
    would need 11.8 GB per head here; the chunked form runs it in a few hundred MB of attention scratch"""
    import torch
    from objectpermanence_amd import ModelsFactory, l1_mean
    cfg = {"boxes_features_dim": 256, "num_attention_heads": 2, "num_attention_layers": 1, "num_lstm_layers": 1,
           "lstm_hidden_dim": 512}
    boxes, labels = synth.make_batch(0, 4, 300)
    held = torch.cuda.memory_allocated()          # what earlier tests of the session still hold is not this step's
    x = torch.from_numpy(np.tile(synth.boxes5(boxes), (32, 1, 1, 1))).cuda()
    lab = torch.from_numpy(np.tile(labels, (32, 1, 1))).cuda()
    m = ModelsFactory.get_model("transformer_lstm", cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.transformer_lstm_synth_params(cfg).items()})
    m.to("cuda:0").train(True)
    torch.cuda.reset_peak_memory_stats()
    loss = l1_mean(m(x), lab)
    loss.backward()
    torch.cuda.synchronize()
    assert np.isfinite(float(loss.detach()))
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())
    assert torch.cuda.max_memory_allocated() - held < 6 * 2 ** 30


# ---- TransformerLstm's TRAIN mode at the reference's dropout 0.1 -------------------------------------

@pytest.mark.gpu
@pytest.mark.parametrize("stack_engine", ["auto", "chain"], indirect=True)
@pytest.mark.parametrize("tag", ["tiny", "real", "heads4"])
def test_transformer_train_mode_with_the_references_dropout_masks(golden_dir, tag, stack_engine):
    """One training step of the reference's own TransformerLstm under model.train() (training_main.py:167: dropout 0.1 live at
    four sites per encoder layer, learned_models.py:166-168) with the masks it drew recorded
    (tests/golden/transformer_dropout_train.npz, oracle/gen_golden.py gen_transformer_dropout).  The HIP encoder is fed the same
    masks through its test-only table (opseq_encoder_test_masks_set) and must reproduce y, the loss and every gradient."""
    import torch
    from objectpermanence_amd import ModelsFactory, _lib, l1_mean
    g = np.load(os.path.join(golden_dir, "transformer_dropout_train.npz"))
    pre = f"{tag}/"
    cfg = json.loads(str(g[pre + "cfg"]))
    n, t = (int(v) for v in g[pre + "shape"])
    boxes, labels = synth.make_batch(0, n, t)
    x = synth.boxes5(boxes)
    m = ModelsFactory.get_model("transformer_lstm", cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in PARAMS["transformer_lstm"](cfg).items()})
    m.to("cuda:0").train(True)
    assert m.dropout == 0.1                                   # the reference's nn.TransformerEncoderLayer default, untouched
    masks = {}
    for li in range(cfg["num_attention_layers"]):
        per_site = []
        for site in range(4):
            shape = tuple(int(v) for v in g[pre + f"mask_shape/{li}/{site}"])
            bits = np.unpackbits(g[pre + f"mask/{li}/{site}"])[:int(np.prod(shape))]
            per_site.append(torch.from_numpy(bits.astype(np.uint8)).cuda().contiguous())
        masks[li] = tuple(per_site)
        assert 0.85 < float(per_site[2].float().mean()) < 0.95
    m._test_dropout_masks = masks
    try:
        y = m(torch.from_numpy(x).cuda())
        loss = l1_mean(y, torch.from_numpy(labels).cuda())
        loss.backward()
        torch.cuda.synchronize()
    finally:
        m._test_dropout_masks = None
        _lib.check(_lib.load().opseq_encoder_test_masks_clear(), "opseq_encoder_test_masks_clear")
    assert np.abs(y.detach().cpu().numpy() - g[pre + "y"]).max() < 5e-5
    _check(g, pre, float(loss.detach()), {k: p.grad.cpu().numpy() for k, p in m.named_parameters()}, 5e-4)
    # ... and the masks mattered: the same step at p = 0 gives another loss
    m.dropout = 0.0
    loss0 = l1_mean(m(torch.from_numpy(x).cuda()), torch.from_numpy(labels).cuda())
    assert abs(float(loss0) - float(g[pre + "loss"])) > 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("ntok,nslots,F", [(300, 1, 256), (1234, 15, 256), (77, 15, 40), (9600, 15, 256)])
def test_slot_embed_weight_gradient_with_workspace_matches_torch_and_the_column_walk(ntok, nslots, F):
    """opseq_slot_embed_relu_bwd_ws_f32 (rows read as they lie, per-workgroup partial sums added in order) against torch's fp64
    gradient of relu(x W^T) and against opseq_slot_embed_relu_bwd_f32 (one workgroup per feature, strided reads); run to run bit-identical"""
    import torch
    from objectpermanence_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(ntok + F)
    x = torch.rand((ntok, 15, 5), generator=g)
    W = (torch.rand((F, 5), generator=g) - 0.5)
    dout = torch.randn((ntok, nslots, F), generator=g)
    xd, Wd, dd = x.cuda(), W.cuda(), dout.cuda()
    out = torch.empty((ntok, nslots * F), device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.opseq_slot_embed_relu_f32(xd.data_ptr(), Wd.data_ptr(), out.data_ptr(), ntok, nslots, F, st), "fwd")
    ws = torch.empty(int(lib.opseq_slot_embed_bwd_workspace_bytes(ntok, nslots, F)), dtype=torch.uint8, device="cuda")
    got = [torch.empty((F, 5), device="cuda") for _ in range(3)]
    for k in (0, 1):
        _lib.check(lib.opseq_slot_embed_relu_bwd_ws_f32(xd.data_ptr(), out.data_ptr(), dd.data_ptr(), got[k].data_ptr(), ntok, nslots, F,
                                                        ws.data_ptr(), ws.numel(), st), "bwd ws")
    _lib.check(lib.opseq_slot_embed_relu_bwd_f32(xd.data_ptr(), out.data_ptr(), dd.data_ptr(), got[2].data_ptr(), ntok, nslots, F, st), "bwd")
    torch.cuda.synchronize()
    W64 = W.double().requires_grad_(True)
    ref_out = torch.relu(x[:, :nslots].double() @ W64.t())
    (ref_out * dout.double()).sum().backward()
    ref = W64.grad
    scale = float(ref.abs().max())
    assert torch.equal(got[0], got[1])
    assert (got[0].cpu().double() - ref).abs().max() <= 2e-5 * scale
    assert (got[0] - got[2]).abs().max().item() <= 2e-5 * scale
    assert lib.opseq_slot_embed_relu_bwd_ws_f32(xd.data_ptr(), out.data_ptr(), dd.data_ptr(), got[0].data_ptr(), ntok, nslots, F,
                                                ws.data_ptr(), 16, st) == -3          # OPNET_EWORKSPACE
