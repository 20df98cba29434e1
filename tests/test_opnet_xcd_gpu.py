"""GPU parity tests of the per-XCD persistent OPNet forward (csrc/opnet_xcd_kernels.hip, opnet_xcd_forward_f32)
against the reference's goldens, the fp64 oracle and the step-launch form.  `pytest -m gpu` on the MI355X box."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import opnet_oracle as oo
from oracle import synth

pytestmark = pytest.mark.gpu

REAL_CFG = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}
TOL_Y = 2e-5          # fp32 through 300 recurrent steps, as tests/test_opnet_gpu.py
TOL_LOGITS = 1e-4


def _model(xcd, device="cuda:0"):
    from objectpermanence_amd import ModelsFactory
    m = ModelsFactory.get_model("opnet", REAL_CFG)
    params = synth.opnet_synth_params(REAL_CFG)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    m.eval()
    m.use_xcd = "1" if xcd else "0"
    return m.to(device), params


def _run(m, boxes):
    with torch.no_grad():
        y, lg = m(torch.from_numpy(boxes).to("cuda:0"))
    torch.cuda.synchronize()
    if m.use_xcd == "1":
        for key, st in m.xcd_status().items():
            assert st[0] == 0, f"persistent launch {key} aborted: block {st[1]} phase {st[2]}"
    return y.cpu().numpy(), lg.cpu().numpy()


def test_persistent_forward_matches_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "opnet_real.npz"))
    assert json.loads(str(g["cfg"])) == REAL_CFG
    n, t = int(g["n_clips"]), int(g["t_frames"])
    boxes, _ = synth.make_batch(0, n, t)
    m, _ = _model(True)
    y, lg = _run(m, boxes)
    assert y.shape == g["y"].shape and lg.shape == g["logits"].shape
    assert np.abs(y - g["y"]).max() < TOL_Y
    assert np.abs(y[:, -5:] - g["y"][:, -5:]).max() < TOL_Y
    assert np.abs(lg - g["logits"]).max() < TOL_LOGITS
    px, px_ref = oo.postprocess_to_pixels(y), oo.postprocess_to_pixels(g["y"])
    assert (px != px_ref).sum() <= 6 and np.abs(px - px_ref).max() <= 1


# one group on one XCD (exposed exchange), ragged last group, one group on every XCD, uneven groups per XCD,
# two and three groups per XCD (the overlapped ring), single frame
@pytest.mark.parametrize("B,T", [(1, 1), (3, 7), (16, 5), (17, 9), (128, 6), (150, 11), (256, 8), (300, 5), (384, 4)])
def test_persistent_forward_matches_oracle_ragged(B, T):
    boxes, _ = synth.make_batch(100, B, T)
    m, params = _model(True)
    y, lg = _run(m, boxes)
    y_ref, lg_ref = oo.opnet_forward(boxes, params, dtype=np.float64)
    assert np.isfinite(y).all()
    assert np.abs(y - y_ref).max() < TOL_Y
    assert np.abs(lg - lg_ref).max() < TOL_LOGITS


@pytest.mark.parametrize("ho", ["0", "1"])
@pytest.mark.parametrize("B,T", [(5, 1), (16, 2), (40, 3), (130, 9), (256, 7), (400, 12), (600, 5)])
def test_both_head_forms_match_oracle(monkeypatch, ho, B, T):
    """The library picks the "head once" form (selection head on one wave per XCD and phase, LSTM2 one more step behind,
    frames_boxes exchanged like h) when every XCD carries >= 3 groups, else the every-CU-computes-the-head form;
    OPNET_XCD_HO forces either one for every shape - both must match the oracle on all of them, 1-2 frame clips included."""
    monkeypatch.setenv("OPNET_XCD_HO", ho)
    boxes, _ = synth.make_batch(200, B, T)
    m, params = _model(True)
    y, lg = _run(m, boxes)
    y_ref, lg_ref = oo.opnet_forward(boxes, params, dtype=np.float64)
    assert np.isfinite(y).all()
    assert np.abs(y - y_ref).max() < TOL_Y
    assert np.abs(lg - lg_ref).max() < TOL_LOGITS


def test_persistent_forward_full_size_matches_step_launch_form_and_is_deterministic():
    """BASELINE shape on every XCD: 8 x 32 clips x 300 frames.  The persistent form sums K in a different order than
    the step launches -> rounding-level agreement; and it must reproduce ITSELF bit for bit from run to run (a stale
    hand-off would not)."""
    boxes, _ = synth.make_batch(0, 384, 300)        # 3 groups per XCD: the "head once" form
    m, _ = _model(True)
    y, lg = _run(m, boxes)
    m0, _ = _model(False)
    y0, lg0 = _run(m0, boxes)
    assert np.abs(y - y0).max() < 1e-5
    assert np.abs(lg - lg0).max() < 5e-5
    for _ in range(3):
        y2, lg2 = _run(m, boxes)
        assert np.array_equal(y2, y) and np.array_equal(lg2, lg)


def test_bench_shape_matches_c_oracle_on_every_clip():
    """The driver's bench shape - twenty 32-clip requests = ONE 640-clip x 300-frame persistent launch (5 groups per XCD,
    head-once form) - against the C port of the reference algorithm (oracle/opnet_oracle.c) on ALL 640 distinct clips,
    not on a sample: float boxes within TOL_Y, int32 pixel boxes equal up to rare 1-px truncation flips."""
    from oracle import c_oracle
    boxes, _ = synth.make_batch(0, 640, 300)
    m, params = _model(True)
    y, lg = _run(m, boxes)
    y_ref, lg_ref = c_oracle.opnet_forward(boxes, params, c_oracle.usable_cores())
    assert y.shape == y_ref.shape == (640, 300, 4)
    err = np.abs(y - y_ref).reshape(640, -1).max(axis=1)
    assert err.max() < TOL_Y, f"worst clip {int(err.argmax())}: {err.max()}"
    assert np.abs(lg - lg_ref).max() < TOL_LOGITS
    px, px_ref = oo.postprocess_to_pixels(y), oo.postprocess_to_pixels(y_ref)
    assert (px != px_ref).mean() < 1e-3 and np.abs(px - px_ref).max() <= 1


@pytest.mark.parametrize("B,T", [(40, 9), (150, 7), (300, 12), (640, 5)])
def test_ring_layout_and_full_history_layout_agree(monkeypatch, B, T):
    """OPNET_XCD_RING=1 (default): h1 / h2 / frames_boxes live in 4-slot rings and y leaves the launch as per-CU partials;
    =0: round 2's full histories + the output-head kernel.  The recurrences are the same instructions (logits bit-identical);
    y differs only in the summation order of its 512-term dot product; both match the oracle."""
    boxes, _ = synth.make_batch(321, B, T)
    outs = {}
    for ring in ("1", "0"):
        monkeypatch.setenv("OPNET_XCD_RING", ring)
        m, params = _model(True)
        outs[ring] = _run(m, boxes)
    y_ref, lg_ref = oo.opnet_forward(boxes, params, dtype=np.float64)
    for ring in ("1", "0"):
        assert np.abs(outs[ring][0] - y_ref).max() < TOL_Y and np.abs(outs[ring][1] - lg_ref).max() < TOL_LOGITS
    assert np.array_equal(outs["1"][1], outs["0"][1])
    assert np.abs(outs["1"][0] - outs["0"][0]).max() < 2e-6


def test_persistent_forward_chunks_large_batches():
    """more clips than one launch carries (opnet_xcd_max_batch) are run as several chained launches"""
    from objectpermanence_amd import _lib
    step = int(_lib.load().opnet_xcd_max_batch())
    B = step + 40
    boxes, _ = synth.make_batch(3, B, 3)
    m, params = _model(True)
    y, lg = _run(m, boxes)
    y_ref, lg_ref = oo.opnet_forward(boxes, params, dtype=np.float64)
    assert np.abs(y - y_ref).max() < TOL_Y and np.abs(lg - lg_ref).max() < TOL_LOGITS


@pytest.mark.parametrize("B", [256, 512])
def test_persistent_forward_under_concurrent_load_on_other_streams(B):
    """Uneven load: a second stream keeps streaming kernels running while the persistent launch is resident, and two
    persistent forwards are issued from different streams (the library chains them through an event).  256 clips = the
    every-CU-head form, 512 = the head-once form."""
    boxes, _ = synth.make_batch(9, B, 40)
    m, _ = _model(True)
    y_ref, lg_ref = _run(m, boxes)
    xb = torch.from_numpy(boxes).to("cuda:0")
    big = torch.empty(64 << 20, dtype=torch.float32, device="cuda:0")
    s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    outs = []
    with torch.no_grad():
        for it in range(3):
            with torch.cuda.stream(s3):
                for _ in range(20):
                    big.mul_(1.0001)
            with torch.cuda.stream(s1):
                outs.append(m(xb))
            with torch.cuda.stream(s2):
                outs.append(m(xb))
    torch.cuda.synchronize()
    for key, st in m.xcd_status().items():
        assert st[0] == 0
    for y, lg in outs[-2:]:
        assert np.array_equal(y.cpu().numpy(), y_ref) and np.array_equal(lg.cpu().numpy(), lg_ref)


@pytest.mark.parametrize("B", [256, 512])
def test_write_through_protocol_gives_the_same_bits(monkeypatch, B):
    """OPNET_XCD_SAFE=1 forces the placement-independent hand-off (write-through stores, every read across the fabric)
    that the kernel falls back to when a group's workgroups are not on one XCD; same arithmetic, same bits."""
    boxes, _ = synth.make_batch(21, B, 30)
    m, _ = _model(True)
    y, lg = _run(m, boxes)
    monkeypatch.setenv("OPNET_XCD_SAFE", "1")
    y2, lg2 = _run(m, boxes)
    assert np.array_equal(y2, y) and np.array_equal(lg2, lg)


def test_workgroups_of_a_group_share_an_xcd():
    """placement is for speed only (block b -> XCD b % 8 is observed, not promised): report it"""
    boxes, _ = synth.make_batch(1, 256, 2)
    m, _ = _model(True)
    _run(m, boxes)
    ws = next(iter(m._xws.values()))
    xcc = ws[32:32 + 4 * 256].view(torch.int32).cpu().numpy()
    same = all(len(set(xcc[x::8].tolist())) == 1 for x in range(8))
    not_local = int(ws[12:16].view(torch.int32).item())
    print("XCC ids of blocks 0..15:", xcc[:16].tolist(), "groups XCD-local:", same, "| groups on the write-through path:", not_local)
    assert (not_local == 0) == same
    assert set(xcc.tolist()) <= set(range(8))


@pytest.mark.parametrize("B", [256, 512])
def test_launch_that_cannot_complete_aborts_instead_of_hanging(monkeypatch, B):
    """Every spin in the persistent kernel is bounded: with the flag publication switched off (OPNET_XCD_DEBUG bit 3, a
    tools-only switch) no consumer ever sees its producers; after 1.5 s the pollers raise the abort word, every wave
    leaves, and the output head fills y with NaN - the status words say which block gave up."""
    import time
    boxes, _ = synth.make_batch(4, B, 6)
    m, _ = _model(True)
    monkeypatch.setenv("OPNET_XCD_DEBUG", "8")
    t0 = time.perf_counter()
    with torch.no_grad():
        y, _ = m(torch.from_numpy(boxes).to("cuda:0"))
    torch.cuda.synchronize()
    assert time.perf_counter() - t0 < 20.0
    st = next(iter(m.xcd_status().values()))
    assert st[0] != 0 and 0 <= st[1] < 256
    assert torch.isnan(y).all()
    monkeypatch.delenv("OPNET_XCD_DEBUG")
    y2, _ = _run(m, boxes)                  # and the next launch is healthy again
    assert np.isfinite(y2).all()


def test_requests_are_read_where_they_lie():
    """opnet_xcd_forward_multi_f32 (OPNet.forward_requests, ReasonerServer.flush): one launch over several request tensors
    = the launch over their concatenation, bit for bit."""
    from objectpermanence_amd.serving import ReasonerServer
    m, _ = _model(True)
    sizes = [32, 7, 100, 1, 48]
    boxes, _ = synth.make_batch(77, sum(sizes), 25)
    xs = torch.from_numpy(boxes).to("cuda:0")
    parts, lo = [], 0
    for n in sizes:
        parts.append(xs[lo:lo + n].clone())
        lo += n
    with torch.no_grad():
        y_cat, lg_cat = m(xs)
        y_req, lg_req = m.forward_requests(parts)
    torch.cuda.synchronize()
    assert torch.equal(y_cat, y_req) and torch.equal(lg_cat, lg_req)
    server = ReasonerServer(m, "opnet", max_clips=1024, concat=False)
    handles = [server.submit(p) for p in parts]
    server.flush()
    lo = 0
    for n, h in zip(sizes, handles):
        y, lg = h.result()
        assert torch.equal(y, y_cat[lo:lo + n]) and torch.equal(lg, lg_cat[lo:lo + n])
        lo += n
    assert server.forwards == 1 and server.clips == sum(sizes)
