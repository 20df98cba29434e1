"""The oracle's pin, made permanent: `oracle/gen_golden.py` imports the reference's own classes from /root/reference and writes
the fixtures under tests/golden/.  Wherever the reference is present (the build container) this test re-runs the generator into
a temporary directory and asserts that EVERY committed fixture is what the reference produces today - arrays bit for bit, text
files character for character.  On the GPU box (no /root/reference) it is skipped: the fixtures travel, the reference cannot."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("OPNET_REFERENCE", "/root/reference")


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "baselines")), reason="the reference is not on this machine")
def test_generator_output_equals_committed_fixtures(tmp_path, golden_dir):
    # the one fixture the generator CONSUMES: weights trained on the MI355X by this repo (data, not reference output)
    shutil.copy(os.path.join(golden_dir, "opnet_trained_fp16.npz"), tmp_path / "opnet_trained_fp16.npz")
    env = dict(os.environ, OPNET_GOLDEN_OUT=str(tmp_path), OPNET_REFERENCE=REF)
    r = subprocess.run([sys.executable, os.path.join(REPO, "oracle", "gen_golden.py")], env=env, cwd=REPO, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    committed = sorted(f for f in os.listdir(golden_dir) if not f.startswith("."))
    assert sorted(os.listdir(tmp_path)) == committed                      # nothing missing, nothing the generator does not make
    for name in committed:
        a, b = os.path.join(golden_dir, name), str(tmp_path / name)
        if name.endswith(".npz"):
            ga, gb = np.load(a, allow_pickle=False), np.load(b, allow_pickle=False)
            assert sorted(ga.files) == sorted(gb.files), name
            for k in ga.files:
                assert ga[k].dtype == gb[k].dtype and ga[k].shape == gb[k].shape, (name, k)
                assert np.array_equal(ga[k], gb[k], equal_nan=ga[k].dtype.kind == "f"), (name, k)
        else:
            assert open(a).read() == open(b).read(), name
