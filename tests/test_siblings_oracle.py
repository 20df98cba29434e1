"""Pin the numpy restatements of the sibling reasoners (BaselineLstm, NonLinearLstm, OPNetLstmMlp,
TransformerLstm) against outputs of the reference's own classes (tests/golden/siblings.npz)."""
import json
import os

import numpy as np
import pytest

from oracle import opnet_oracle as oo, synth


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "siblings.npz"))


def _case(gold, name, tag):
    cfg = json.loads(str(gold[f"{name}/{tag}/cfg"]))
    n, t = (int(v) for v in gold[f"{name}/{tag}/shape"])
    boxes, _ = synth.make_batch(0, n, t)
    return cfg, boxes, gold[f"{name}/{tag}/y"]


@pytest.mark.parametrize("tag", ["tiny", "real"])
def test_baseline_lstm(gold, tag):
    cfg, boxes, y_ref = _case(gold, "baseline_lstm", tag)
    y = oo.baseline_lstm_forward(synth.boxes5(boxes), synth.baseline_lstm_synth_params(cfg))
    assert np.abs(y - y_ref).max() < 1e-5 and y_ref.std() > 0.1


@pytest.mark.parametrize("tag", ["tiny", "real"])
def test_non_linear_lstm(gold, tag):
    cfg, boxes, y_ref = _case(gold, "non_linear_lstm", tag)
    y = oo.non_linear_lstm_forward(synth.boxes5(boxes), synth.non_linear_lstm_synth_params(cfg))
    assert np.abs(y - y_ref).max() < 1e-5 and y_ref.std() > 0.1


@pytest.mark.parametrize("tag", ["tiny", "real"])
def test_opnet_lstm_mlp(gold, tag):
    cfg, boxes, y_ref = _case(gold, "opnet_lstm_mlp", tag)
    y, lg = oo.opnet_lstm_mlp_forward(boxes, synth.opnet_lstm_mlp_synth_params(cfg))
    assert np.abs(y - y_ref).max() < 1e-5
    assert np.abs(lg - gold[f"opnet_lstm_mlp/{tag}/logits"]).max() < 5e-5


@pytest.mark.parametrize("tag", ["tiny", "real_b1", "real_b2", "heads4_b1"])
def test_transformer_lstm_slot0_path_equals_reference_full_evaluation(gold, tag):
    cfg, boxes, y_ref = _case(gold, "transformer_lstm", tag)
    y = oo.transformer_lstm_forward(synth.boxes5(boxes), synth.transformer_lstm_synth_params(cfg), cfg)
    assert np.abs(y - y_ref).max() < 2e-5
    assert y_ref.std() > 0.1


def test_transformer_output_depends_on_batch_composition(gold):
    """attention spans all B*T frames of the minibatch - clip 0 alone != clip 0 in a
    batch of two; a faithful implementation must reproduce exactly that."""
    y1 = gold["transformer_lstm/real_b1/y"][0]
    y2 = gold["transformer_lstm/real_b2/y"][0]
    assert np.abs(y1 - y2).max() > 1e-2
