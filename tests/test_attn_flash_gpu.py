"""The flash-style training attention (csrc/attn_train_kernels.hip: scores in registers, forward keeps (row max, 1 / row sum),
the backward recomputes score tiles in a query-stationary and a key-stationary pass) against the chunked GEMM form it replaces
(OPSEQ_ATTN_FLASH=0, itself pinned to the reference's autograd and recorded dropout masks in tests/test_siblings_train.py),
for every head size the kernels are built for, with and without dropout, split sweeps and two fragments per wave."""
import numpy as np
import pytest

from oracle import synth

pytestmark = pytest.mark.gpu


def _step(cfg, B, T, p_drop, seed=21):
    import torch
    from objectpermanence_amd import ModelsFactory, l1_mean
    boxes, labels = synth.make_batch(seed, B, T)
    x, lab = torch.from_numpy(synth.boxes5(boxes)).cuda(), torch.from_numpy(labels).cuda()
    m = ModelsFactory.get_model("transformer_lstm", cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.transformer_lstm_synth_params(cfg).items()})
    m.to("cuda:0").train(True)
    m.dropout = p_drop
    m._calls = 3
    y = m(x)
    loss = l1_mean(y, lab)
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.detach()), y.detach().cpu().numpy(), {k: p.grad.cpu().numpy() for k, p in m.named_parameters()}


def _cfg(E, nhead):
    return {"boxes_features_dim": E, "num_attention_heads": nhead, "num_attention_layers": 2, "num_lstm_layers": 2,
            "lstm_hidden_dim": 48}


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
@pytest.mark.parametrize("B,T,E,nhead", [(3, 37, 64, 4), (2, 50, 64, 2), (1, 70, 128, 2), (2, 41, 256, 2), (5, 60, 256, 4)])
def test_flash_attention_training_matches_the_chunked_form(monkeypatch, B, T, E, nhead, p_drop):
    """head sizes 16 / 32 / 64 / 128 (and 64 at 4 heads), ragged S; the dropout multipliers are keyed by the element's global
    index in both forms, so the SAME masks apply"""
    monkeypatch.setenv("OPSEQ_ATTN_FLASH", "0")
    l_ref, y_ref, g_ref = _step(_cfg(E, nhead), B, T, p_drop)
    monkeypatch.setenv("OPSEQ_ATTN_FLASH", "1")
    l, y, g = _step(_cfg(E, nhead), B, T, p_drop)
    assert np.abs(y - y_ref).max() < 2e-5
    assert l == pytest.approx(l_ref, abs=2e-6)
    _close(g, g_ref)


def _close(g, g_ref, fro=5e-4, worst=3e-3):
    """Two forwards that differ in the last bits put a handful of the 600 k FFN pre-activations of a step on the other side of the
    ReLU (measured: about one per layer and run), which moves single rows of a weight gradient by ~1e-4 of max|g| - so the gradients
    are held together in the Frobenius norm (5e-4: a flip in the upper layer moves everything below it; an indexing error gives O(1)) with a loose element-wise bound; the reference's own
    goldens bound the flash path element-wise in tests/test_siblings_train.py"""
    for k in g_ref:
        a, b = g[k].astype(np.float64), g_ref[k].astype(np.float64)
        assert np.isfinite(a).all(), k
        assert np.sqrt(((a - b) ** 2).sum()) <= fro * max(1e-6, np.sqrt((b ** 2).sum())), k
        assert np.abs(a - b).max() <= worst * max(1e-2, np.abs(b).max()), k


@pytest.mark.parametrize("E,nhead", [(128, 2), (256, 2)])
def test_flash_attention_split_sweeps_and_fragment_counts_agree(monkeypatch, E, nhead):
    """S = 17 x 64 = 1088 tokens: the streaming sweep cut in 2 / 3 slices (partials summed in slice order) and two stationary
    fragments per wave give the single-sweep gradients up to the summation order; every form is bit-reproducible run to run"""
    cfg = _cfg(E, nhead)
    monkeypatch.delenv("OPSEQ_ATTN_ZS", raising=False)
    monkeypatch.setenv("OPSEQ_ATTN_ZS", "1")
    monkeypatch.setenv("OPSEQ_ATTN_AF", "1")
    l_ref, y_ref, g_ref = _step(cfg, 17, 64, 0.1)
    l_again, y_again, g_again = _step(cfg, 17, 64, 0.1)
    assert l_again == l_ref and np.array_equal(y_again, y_ref) and all(np.array_equal(g_again[k], g_ref[k]) for k in g_ref)
    for zs, af in ((2, 1), (3, 1), (1, 2), (2, 2)):
        monkeypatch.setenv("OPSEQ_ATTN_ZS", str(zs))
        monkeypatch.setenv("OPSEQ_ATTN_AF", str(af))
        l, y, g = _step(cfg, 17, 64, 0.1)
        assert np.abs(y - y_ref).max() < 1e-5, (zs, af)
        _close(g, g_ref)
        l2, y2, g2 = _step(cfg, 17, 64, 0.1)
        assert l2 == l and np.array_equal(y2, y) and all(np.array_equal(g2[k], g[k]) for k in g), (zs, af)
