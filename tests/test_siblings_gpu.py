"""GPU parity of the sibling reasoners (BaselineLstm, NonLinearLstm, OPNetLstmMlp, TransformerLstm)
against outputs of the reference's own classes (tests/golden/siblings.npz) and the numpy oracle."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import opnet_oracle as oo, synth

pytestmark = pytest.mark.gpu

PARAMS = {"baseline_lstm": synth.baseline_lstm_synth_params, "non_linear_lstm": synth.non_linear_lstm_synth_params,
          "opnet_lstm_mlp": synth.opnet_lstm_mlp_synth_params, "transformer_lstm": synth.transformer_lstm_synth_params}


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "siblings.npz"))


def _model(name, cfg, engine="auto"):
    """engine: "auto" = the product's choice (H = 512 stacks: the persistent launch of 4-clip groups, csrc/seq_xcd_kernels.hip,
    below 64 clips and of 16-clip groups, csrc/seq_xcdt_kernels.hip, from there on), "latency" / "throughput" = one of the two
    whatever the batch, "chain" = one launch per time step (csrc/seq_kernels.hip)"""
    from objectpermanence_amd import ModelsFactory
    m = ModelsFactory.get_model(name, cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in PARAMS[name](cfg).items()})
    if hasattr(m, "_runner"):
        if engine == "chain":
            m._runner.use_xcd = "0"
        elif engine == "latency":
            m._runner.use_xcdt = "0"
        elif engine == "throughput":
            m._runner.use_xcdt = "1"
    return m.eval().to("cuda:0")


def _persistent(m, form="latency"):
    return getattr(getattr(m, "_runner", None), "xcdt_launches" if form == "throughput" else "xcd_launches", 0)


def _run(m, x):
    with torch.no_grad():
        out = m(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    return out


CASES = [("baseline_lstm", "tiny"), ("baseline_lstm", "real"), ("non_linear_lstm", "tiny"), ("non_linear_lstm", "real"),
         ("opnet_lstm_mlp", "tiny"), ("opnet_lstm_mlp", "real"), ("transformer_lstm", "tiny"),
         ("transformer_lstm", "real_b1"), ("transformer_lstm", "real_b2"), ("transformer_lstm", "heads4_b1")]


@pytest.mark.parametrize("engine", ["auto", "chain", "throughput"])
@pytest.mark.parametrize("name,tag", CASES)
def test_matches_reference_golden(gold, name, tag, engine):
    """the three engines of the stacked LSTM against the reference's own outputs: the persistent launch of 4-clip groups (what
    "auto" picks for the goldens' few clips at the real configs: H = 512), the launch-per-step chain, and the persistent launch of
    16-clip groups (what batches of 64 clips or more run)"""
    cfg = json.loads(str(gold[f"{name}/{tag}/cfg"]))
    n, t = (int(v) for v in gold[f"{name}/{tag}/shape"])
    boxes, _ = synth.make_batch(0, n, t)
    x = boxes if name == "opnet_lstm_mlp" else synth.boxes5(boxes)
    m = _model(name, cfg, engine)
    out = _run(m, x)
    real_stack = name != "opnet_lstm_mlp" and tag != "tiny"
    if engine == "throughput" and not real_stack:
        pytest.skip("the throughput form is built for the H = 512 stacks")
    assert _persistent(m) == (1 if (engine == "auto" and real_stack) else 0)
    assert _persistent(m, "throughput") == (1 if engine == "throughput" else 0)
    if hasattr(m, "_runner"):
        assert m._runner._monitor.verify() == 0
    y = (out[0] if isinstance(out, tuple) else out).cpu().numpy()
    y_ref = gold[f"{name}/{tag}/y"]
    assert y.shape == y_ref.shape
    assert np.abs(y - y_ref).max() < 3e-5, float(np.abs(y - y_ref).max())
    if isinstance(out, tuple):
        assert np.abs(out[1].cpu().numpy() - gold[f"{name}/{tag}/logits"]).max() < 1e-4


def test_transformer_batch_coupling_and_ragged_vs_oracle():
    """S = B*T attention: ragged S (not a multiple of 16/64), several clips, vs the fp64 oracle."""
    cfg = {"boxes_features_dim": 256, "num_attention_heads": 2, "num_attention_layers": 2,
           "num_lstm_layers": 2, "lstm_hidden_dim": 512}
    boxes, _ = synth.make_batch(40, 3, 37)     # S = 111
    x = synth.boxes5(boxes)
    m = _model("transformer_lstm", cfg)
    y = _run(m, x).cpu().numpy()
    y_ref = oo.transformer_lstm_forward(x, PARAMS["transformer_lstm"](cfg), cfg)
    assert np.abs(y - y_ref).max() < 3e-5
    y_single = _run(m, x[:1]).cpu().numpy()
    assert np.abs(y_single[0] - y[0]).max() > 1e-3      # clip 0 alone != clip 0 in the batch (reference quirk)


@pytest.mark.parametrize("name,B,T", [("baseline_lstm", 33, 9), ("non_linear_lstm", 5, 4), ("opnet_lstm_mlp", 40, 7),
                                      ("baseline_lstm_wide", 6, 5), ("baseline_lstm_wide", 37, 4)])
def test_ragged_batches_vs_oracle(name, B, T):
    cfgs = {"baseline_lstm": {"videos_hidden_dim": 512},
            # a wave's K slice (16 hexadecets) exceeds the register chunk: weight fragments are streamed
            "baseline_lstm_wide": {"videos_hidden_dim": 1024},
            "non_linear_lstm": {"boxes_features_dim": 32, "videos_hidden_dim": 64},
            "opnet_lstm_mlp": {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}}
    cfg = cfgs[name]
    name = name.replace("_wide", "")
    boxes, _ = synth.make_batch(300, B, T)
    p = PARAMS[name](cfg)
    if name == "opnet_lstm_mlp":
        y_ref, _ = oo.opnet_lstm_mlp_forward(boxes, p)
        y = _run(_model(name, cfg), boxes)[0].cpu().numpy()
    else:
        fn = oo.baseline_lstm_forward if name == "baseline_lstm" else oo.non_linear_lstm_forward
        y_ref = fn(synth.boxes5(boxes), p)
        y = _run(_model(name, cfg), synth.boxes5(boxes)).cpu().numpy()
    assert np.abs(y - y_ref).max() < 3e-5


@pytest.mark.parametrize("S,E,nhead", [
    (300, 256, 2), (300, 256, 4),        # one clip, the JSON's 2 heads and BASELINE.json's 4 heads
    (77, 32, 2), (130, 64, 2),           # head sizes 16 and 32, ragged last key tile
    (200, 96, 2),                        # head size 48: not a power of two -> the direct (un-staged) kernel
    (8400, 256, 4),                      # 28 clips: two query fragments per wave (>= 256 workgroups)
    (16500, 256, 2),
    (9600, 256, 2), (5000, 256, 4),      # workgroup counts that trigger the key split
])
def test_attention_core_matches_torch(S, E, nhead):
    """opseq_attention_f32 (LDS-DMA flash kernel) vs softmax(q k^T / sqrt(hd)) v in torch on the CPU"""
    from objectpermanence_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(S + E + nhead)
    qkv = torch.randn((S, 3 * E), generator=g) * 1.5
    x = qkv.cuda()
    out = torch.empty((S, E), device="cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    ws = torch.empty(lib.opseq_attention_workspace_bytes(S, E, nhead), dtype=torch.uint8, device="cuda:0")
    _lib.check(lib.opseq_attention_f32(x.data_ptr(), out.data_ptr(), S, E, nhead, ws.data_ptr(), ws.numel(), st), "opseq_attention_f32")
    out1 = torch.empty_like(out)                      # without scratch the kernel must run unsplit and agree
    _lib.check(lib.opseq_attention_f32(x.data_ptr(), out1.data_ptr(), S, E, nhead, None, 0, st), "opseq_attention_f32")
    torch.cuda.synchronize()
    assert (out - out1).abs().max().item() < 5e-5      # fp32 reassociation of the key sum
    dt = torch.float64 if S <= 1000 else torch.float32
    hd = E // nhead
    q, k, v = (qkv[:, j * E:(j + 1) * E].to(dt).view(S, nhead, hd).transpose(0, 1) for j in range(3))
    ref = torch.cat([torch.softmax(q[h] @ k[h].T / hd ** 0.5, dim=-1) @ v[h] for h in range(nhead)], dim=1)
    err = (out.cpu().to(dt) - ref).abs().max().item()
    assert err < (2e-5 if S <= 1000 else 2e-4), err


REAL = {"baseline_lstm": {"videos_hidden_dim": 512},
        "non_linear_lstm": {"boxes_features_dim": 256, "videos_hidden_dim": 512},
        "transformer_lstm": {"boxes_features_dim": 256, "num_attention_heads": 2, "num_attention_layers": 2,
                             "num_lstm_layers": 2, "lstm_hidden_dim": 512}}
ORACLE = {"baseline_lstm": lambda x, p, cfg: oo.baseline_lstm_forward(x, p),
          "non_linear_lstm": lambda x, p, cfg: oo.non_linear_lstm_forward(x, p),
          "transformer_lstm": lambda x, p, cfg: oo.transformer_lstm_forward(x, p, cfg)}


# one clip / a ragged last group / every XCD busy / two groups per XCD (pair) / single frame, per model
@pytest.mark.parametrize("name,B,T", [("baseline_lstm", 1, 1), ("baseline_lstm", 3, 7), ("baseline_lstm", 32, 9),
                                      ("baseline_lstm", 37, 5), ("baseline_lstm", 70, 4), ("baseline_lstm", 256, 3),
                                      ("non_linear_lstm", 1, 2), ("non_linear_lstm", 6, 5), ("non_linear_lstm", 17, 6),
                                      ("non_linear_lstm", 40, 3), ("transformer_lstm", 1, 11), ("transformer_lstm", 2, 8),
                                      ("transformer_lstm", 5, 6), ("transformer_lstm", 19, 4), ("transformer_lstm", 33, 3)])
def test_persistent_stack_matches_oracle_ragged(name, B, T):
    """the persistent launch of the stacked LSTM (csrc/seq_xcd_kernels.hip) on ragged shapes vs the fp64 oracle, and
    against the launch chain on the same input (different summation order: rounding-level agreement)"""
    cfg = REAL[name]
    boxes, _ = synth.make_batch(500, B, T)
    x = synth.boxes5(boxes)
    m = _model(name, cfg, "latency")
    y = _run(m, x).cpu().numpy()
    assert _persistent(m) == 1 and m._runner._monitor.verify() == 0
    y_ref = ORACLE[name](x, PARAMS[name](cfg), cfg)
    assert np.isfinite(y).all() and np.abs(y - y_ref).max() < 3e-5, float(np.abs(y - y_ref).max())
    y_chain = _run(_model(name, cfg, "chain"), x).cpu().numpy()
    assert np.abs(y - y_chain).max() < 1e-5


# one clip / ragged last groups / one group on some XCDs (pairs) only / every XCD (pair) busy / two and three groups per XCD
# (pair), T = 1 and 2 (the lagging layer's start-up), per model
@pytest.mark.parametrize("name,B,T", [("baseline_lstm", 1, 1), ("baseline_lstm", 3, 7), ("baseline_lstm", 37, 5),
                                      ("baseline_lstm", 128, 6), ("baseline_lstm", 200, 5), ("baseline_lstm", 380, 4),
                                      ("non_linear_lstm", 1, 2), ("non_linear_lstm", 17, 6), ("non_linear_lstm", 70, 1),
                                      ("non_linear_lstm", 100, 5), ("transformer_lstm", 1, 11), ("transformer_lstm", 19, 4),
                                      ("transformer_lstm", 64, 3), ("transformer_lstm", 150, 2),
                                      # more clips than one launch carries (1024 / 512): the runner loops over chunks
                                      ("baseline_lstm", 1100, 2), ("non_linear_lstm", 530, 2)])
def test_throughput_stack_matches_oracle_ragged(name, B, T):
    """the persistent launch of 16-clip groups (csrc/seq_xcdt_kernels.hip) on ragged shapes vs the fp64 oracle, and against the
    4-clip form on the same input (different summation order: rounding-level agreement)"""
    cfg = REAL[name]
    boxes, _ = synth.make_batch(500, B, T)
    x = synth.boxes5(boxes)
    m = _model(name, cfg, "throughput")
    y = _run(m, x).cpu().numpy()
    from objectpermanence_amd import _lib
    r = m._runner
    cap = int(_lib.load().opseq_xcdt_max_batch(T, r.L, r.KX, r.H))
    assert _persistent(m, "throughput") == -(-B // cap) and _persistent(m) == 0 and m._runner._monitor.verify() == 0
    y_ref = ORACLE[name](x, PARAMS[name](cfg), cfg)
    assert np.isfinite(y).all() and np.abs(y - y_ref).max() < 3e-5, float(np.abs(y - y_ref).max())
    if B <= 128:
        y_lat = _run(_model(name, cfg, "latency"), x).cpu().numpy()
        assert np.abs(y - y_lat).max() < 1e-5


def test_batches_are_routed_to_the_throughput_form():
    """"auto": below _LstmStackRunner.XCDT_MIN_BATCH clips the 4-clip latency form, from there on the 16-clip throughput form"""
    cfg = REAL["baseline_lstm"]
    m = _model("baseline_lstm", cfg)
    lo = m._runner.XCDT_MIN_BATCH
    for B, form in ((lo - 1, "latency"), (lo, "throughput"), (300, "throughput")):
        before = (_persistent(m), _persistent(m, "throughput"))
        x = synth.boxes5(synth.make_batch(20, B, 3)[0])
        y = _run(m, x).cpu().numpy()
        after = (_persistent(m), _persistent(m, "throughput"))
        assert (after[0] - before[0], after[1] - before[1]) == ((1, 0) if form == "latency" else (0, 1)), (B, form)
        assert np.abs(y - oo.baseline_lstm_forward(x, PARAMS["baseline_lstm"](cfg))).max() < 3e-5


@pytest.mark.parametrize("name,B,T", [("baseline_lstm", 300, 30), ("transformer_lstm", 150, 25)])
def test_throughput_stack_is_deterministic_and_protocol_independent(monkeypatch, name, B, T):
    """run to run and XCD-local stores vs the write-through protocol (OPNET_XCD_SAFE): the same bits - a stale hand-off would
    not reproduce; a clip's result does not depend on which column of which group it rides in"""
    cfg = REAL[name]
    x = synth.boxes5(synth.make_batch(7, B, T)[0])
    m = _model(name, cfg, "throughput")
    y0 = _run(m, x).cpu().numpy()
    for _ in range(3):
        assert np.array_equal(_run(m, x).cpu().numpy(), y0)
    monkeypatch.setenv("OPNET_XCD_SAFE", "1")
    assert np.array_equal(_run(m, x).cpu().numpy(), y0)
    monkeypatch.delenv("OPNET_XCD_SAFE")
    if name == "baseline_lstm":                  # clips are independent there: a reversed batch gives the reversed result
        assert np.array_equal(_run(m, x[::-1].copy()).cpu().numpy(), y0[::-1])


def test_throughput_stack_abort_is_healed(monkeypatch):
    """OPSEQ_XCDT_DEBUG bit 3 switches the publish off: the launch gives up after its bounded wait, y is NaN, and verify_launches
    re-runs the batch on the launch chain into the same tensor"""
    import warnings
    from objectpermanence_amd.launch_monitor import verify_launches
    cfg = REAL["baseline_lstm"]
    x = synth.boxes5(synth.make_batch(9, 70, 6)[0])
    m = _model("baseline_lstm", cfg, "throughput")
    monkeypatch.setenv("OPSEQ_XCDT_DEBUG", "8")
    y = _run(m, x)
    assert torch.isnan(y).all()
    monkeypatch.delenv("OPSEQ_XCDT_DEBUG")
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        assert verify_launches(m) == 1
    y_ref = oo.baseline_lstm_forward(x, PARAMS["baseline_lstm"](cfg))
    assert np.abs(y.cpu().numpy() - y_ref).max() < 3e-5
    assert np.abs(_run(m, x).cpu().numpy() - y_ref).max() < 3e-5      # the next launch is healthy


def test_persistent_stack_is_deterministic_and_protocol_independent(monkeypatch):
    """run to run and XCD-local stores vs the write-through protocol (OPNET_XCD_SAFE): the same bits - a stale hand-off
    would not reproduce"""
    cfg = REAL["transformer_lstm"]
    boxes, _ = synth.make_batch(7, 6, 40)
    x = synth.boxes5(boxes)
    m = _model("transformer_lstm", cfg)
    y0 = _run(m, x).cpu().numpy()
    for _ in range(3):
        assert np.array_equal(_run(m, x).cpu().numpy(), y0)
    monkeypatch.setenv("OPNET_XCD_SAFE", "1")
    assert np.array_equal(_run(m, x).cpu().numpy(), y0)


def test_persistent_stack_abort_is_healed(monkeypatch):
    """OPSEQ_XCD_DEBUG bit 2 switches the cells (the publishers) off: the launch gives up after its bounded wait, y is NaN,
    and verify_launches re-runs the batch on the launch chain into the same tensor"""
    import warnings
    from objectpermanence_amd.launch_monitor import verify_launches
    cfg = REAL["baseline_lstm"]
    boxes, _ = synth.make_batch(9, 5, 6)
    x = synth.boxes5(boxes)
    m = _model("baseline_lstm", cfg)
    monkeypatch.setenv("OPSEQ_XCD_DEBUG", "4")
    y = _run(m, x)
    assert torch.isnan(y).all()
    monkeypatch.delenv("OPSEQ_XCD_DEBUG")
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        assert verify_launches(m) == 1
    y_ref = oo.baseline_lstm_forward(x, PARAMS["baseline_lstm"](cfg))
    assert np.abs(y.cpu().numpy() - y_ref).max() < 3e-5
