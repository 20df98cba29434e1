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


def _model(name, cfg):
    from objectpermanence_amd import ModelsFactory
    m = ModelsFactory.get_model(name, cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in PARAMS[name](cfg).items()})
    return m.eval().to("cuda:0")


def _run(m, x):
    with torch.no_grad():
        out = m(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    return out


CASES = [("baseline_lstm", "tiny"), ("baseline_lstm", "real"), ("non_linear_lstm", "tiny"), ("non_linear_lstm", "real"),
         ("opnet_lstm_mlp", "tiny"), ("opnet_lstm_mlp", "real"), ("transformer_lstm", "tiny"),
         ("transformer_lstm", "real_b1"), ("transformer_lstm", "real_b2"), ("transformer_lstm", "heads4_b1")]


@pytest.mark.parametrize("name,tag", CASES)
def test_matches_reference_golden(gold, name, tag):
    cfg = json.loads(str(gold[f"{name}/{tag}/cfg"]))
    n, t = (int(v) for v in gold[f"{name}/{tag}/shape"])
    boxes, _ = synth.make_batch(0, n, t)
    x = boxes if name == "opnet_lstm_mlp" else synth.boxes5(boxes)
    out = _run(_model(name, cfg), x)
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
