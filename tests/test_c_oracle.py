"""Pin the C restatement (oracle/opnet_oracle.c) against the reference-generated goldens and the
numpy oracle; it is the full-size checker and the timed CPU baseline."""
import json
import os

import numpy as np
import pytest

from oracle import c_oracle, opnet_oracle as oo, synth


@pytest.mark.parametrize("tag", ["tiny", "real"])
def test_c_oracle_matches_reference_golden(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, f"opnet_{tag}.npz"))
    cfg = json.loads(str(g["cfg"]))
    boxes, _ = synth.make_batch(0, int(g["n_clips"]), int(g["t_frames"]))
    y, lg = c_oracle.opnet_forward(boxes, synth.opnet_synth_params(cfg))
    assert np.abs(y - g["y"]).max() < 1e-5
    assert np.abs(lg - g["logits"]).max() < 5e-5


def test_c_oracle_ragged_batch_and_threads():
    cfg = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 32, "videos_hidden_dim": 48}
    p = synth.opnet_synth_params(cfg)
    boxes, _ = synth.make_batch(50, 7, 20)   # 7 clips: one full block of 4 + a ragged one
    y1, lg1 = c_oracle.opnet_forward(boxes, p, n_threads=1)
    y3, lg3 = c_oracle.opnet_forward(boxes, p, n_threads=3)
    assert np.array_equal(y1, y3) and np.array_equal(lg1, lg3)   # thread count never changes bits
    y_ref, lg_ref = oo.opnet_forward(boxes, p, np.float64)
    assert np.abs(y1 - y_ref).max() < 5e-6 and np.abs(lg1 - lg_ref).max() < 5e-5
