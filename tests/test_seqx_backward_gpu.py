"""The stacked LSTM's reverse recurrence as ONE persistent launch (csrc/seq_xcd_bwd_kernels.hip) against the launch chain it
replaces (OPSEQ_XCD_BWD=0: stack_bwd_cell / stack_bwd_gemm, pinned to the reference's autograd in tests/test_siblings_train.py):
one and several 4-clip groups per XCD (pair), ragged last groups, T = 1, both layer counts, and bit-reproducibility."""
import numpy as np
import pytest

from oracle import synth

pytestmark = pytest.mark.gpu

CFG = {"baseline_lstm": {"videos_hidden_dim": 512},
       "non_linear_lstm": {"boxes_features_dim": 256, "videos_hidden_dim": 512},
       "transformer_lstm": {"boxes_features_dim": 256, "num_attention_heads": 2, "num_attention_layers": 1,
                            "num_lstm_layers": 2, "lstm_hidden_dim": 512}}
PARAMS = {"baseline_lstm": synth.baseline_lstm_synth_params, "non_linear_lstm": synth.non_linear_lstm_synth_params,
          "transformer_lstm": synth.transformer_lstm_synth_params}


def _step(name, B, T):
    import torch
    from objectpermanence_amd import ModelsFactory, l1_mean
    boxes, labels = synth.make_batch(7, B, T)
    x, lab = torch.from_numpy(synth.boxes5(boxes)).cuda(), torch.from_numpy(labels).cuda()
    m = ModelsFactory.get_model(name, CFG[name])
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in PARAMS[name](CFG[name]).items()})
    m.to("cuda:0").train(True)
    if name == "transformer_lstm":
        m.dropout = 0.0
    loss = l1_mean(m(x), lab)
    loss.backward()
    torch.cuda.synchronize()
    assert not m.training_step_aborted()
    return float(loss.detach()), {k: p.grad.cpu().numpy() for k, p in m.named_parameters()}


@pytest.mark.parametrize("name,B,T", [("baseline_lstm", 1, 1), ("baseline_lstm", 5, 9), ("baseline_lstm", 32, 12),
                                       ("baseline_lstm", 37, 6), ("baseline_lstm", 70, 5), ("transformer_lstm", 1, 8),
                                       ("transformer_lstm", 3, 1), ("transformer_lstm", 18, 7), ("transformer_lstm", 35, 4),
                                       ("non_linear_lstm", 6, 5)])
def test_persistent_reverse_recurrence_matches_the_launch_chain(monkeypatch, name, B, T):
    monkeypatch.setenv("OPSEQ_XCD_BWD", "0")
    l_ref, g_ref = _step(name, B, T)
    monkeypatch.setenv("OPSEQ_XCD_BWD", "1")
    l, g = _step(name, B, T)
    assert l == l_ref                                   # (the forward is the same launch either way)
    for k in g_ref:
        assert np.isfinite(g[k]).all(), k
        assert np.abs(g[k] - g_ref[k]).max() <= 2e-5 * max(1e-3, np.abs(g_ref[k]).max()), k
    l2, g2 = _step(name, B, T)
    assert all(np.array_equal(g2[k], g[k]) for k in g), "run to run"


def test_persistent_reverse_recurrence_under_the_write_through_protocol(monkeypatch):
    """OPNET_XCD_SAFE=1: every exchange store written through (what a launch whose workgroups do not sit on one XCD per group
    does): the same bits as the XCD-local protocol"""
    l, g = _step("transformer_lstm", 5, 6)
    monkeypatch.setenv("OPNET_XCD_SAFE", "1")
    l2, g2 = _step("transformer_lstm", 5, 6)
    assert l2 == l and all(np.array_equal(g2[k], g[k]) for k in g)
