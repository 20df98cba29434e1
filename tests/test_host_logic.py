"""CPU-side checks: the C ABI library loads and exports every symbol include/opnet_hip.h declares, host
argument validation (no GPU work is launched), and the data-parallel sharding logic over gloo."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from objectpermanence_amd import _lib, build
    build.build()
    return _lib.load()


def test_header_symbols_are_exported():
    from objectpermanence_amd import _lib as binding
    lib = _lib()
    hdr = open(os.path.join(REPO, "include", "opnet_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(op(?:net|seq|det)_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(binding.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/opnet_hip.h but not exported"


def test_persistent_forward_workspace_is_rings_not_histories(monkeypatch):
    """the per-XCD persistent forward keeps h1 / h2 / frames_boxes in 4-slot rings: 0.27 MB of workspace per clip at T = 300
    (packed input + per-CU partials of the output head) against 0.94 MB with the full histories of round 2"""
    lib = _lib()
    ring = lib.opnet_xcd_workspace_bytes(1024, 300, 256, 512)
    monkeypatch.setenv("OPNET_XCD_RING", "0")
    full = lib.opnet_xcd_workspace_bytes(1024, 300, 256, 512)
    assert 0 < ring < 300e6 < 900e6 < full


def test_size_queries_and_validation_without_gpu():
    lib = _lib()
    from objectpermanence_amd import _lib as binding
    assert lib.opnet_hip_abi_version() == binding.ABI_VERSION == 9
    w = lib.opnet_workspace_bytes(32, 300, 256, 512)
    assert w > 32 * 300 * 96 * 4 and lib.opnet_workspace_bytes(64, 300, 256, 512) > w
    assert lib.opnet_workspace_bytes(33, 300, 256, 512) == lib.opnet_workspace_bytes(64, 300, 256, 512)
    p = lib.opnet_packed_weights_bytes(256, 512)
    assert p >= 1_421_056 * 4      # every reference parameter is present (padding only adds)
    assert lib.opnet_workspace_bytes(0, 300, 256, 512) == 0
    assert lib.opnet_packed_weights_bytes(250, 512) == 0
    assert b"multiples of 16" in lib.opnet_last_error()
    # null pointers are rejected before anything is enqueued
    assert lib.opnet_forward_f32(None, None, None, None, None, 0, 1, 1, 16, 16, None) == -1
    assert lib.opnet_pack_weights_f32(None, None, None, None, None, None, None, 0, 16, 16, None) == -1
    plan = ctypes.c_void_p()
    assert lib.opnet_plan_create(ctypes.byref(plan), 4, 10, 16, 32) == 0 and plan.value
    assert lib.opnet_plan_forward(plan, None, None, None, None, None, 0, None) == -1
    lib.opnet_plan_destroy(plan)
    assert lib.opnet_plan_create(ctypes.byref(plan), 4, 10, 10, 32) == -2
    assert lib.opnet_postprocess_iou(None, None, None, None, None, 1, 1, None) == -1


def test_tile_plan_of_the_resident_token_kernels_covers_every_row_once():
    """csrc/ffn_kernels.hip (the encoder's feed-forward block of learned_models.py:166-171 as one kernel): full rounds of 64-row tiles
    over the CUs, the rest cut into equal tiles of 16 / 32 / 48 / 64 rows - host logic, no device call"""
    lib = _lib()
    rng = np.random.default_rng(7)
    cases = [(1, 256), (15, 256), (64, 256), (300, 256), (16384, 256), (16385, 256), (76800, 256), (153600, 256), (76800, 304), (5000, 8)]
    cases += [(int(m), int(c)) for m, c in zip(rng.integers(1, 400000, 200), rng.choice([8, 32, 64, 128, 256, 304], 200))]
    for M, cus in cases:
        n_full, frags, grid = ctypes.c_int(-1), ctypes.c_int(-1), ctypes.c_uint(0)
        assert lib.opseq_ffn_fused_plan(M, cus, ctypes.byref(n_full), ctypes.byref(frags), ctypes.byref(grid)) == 0
        n_full, frags, grid = n_full.value, frags.value, grid.value
        assert 1 <= frags <= 4 and n_full % cus == 0 and n_full * 64 <= M
        tail = grid - n_full
        rem = M - n_full * 64
        assert rem < 64 * cus                                   # the tail is less than one full round
        assert tail * 16 * frags >= rem and (tail == 0) == (rem == 0)
        if tail:
            assert (tail - 1) * 16 * frags < rem                # no empty workgroup
            assert tail <= cus or frags == 4                    # one tail tile per CU unless a CU's share exceeds a full tile
    assert lib.opseq_ffn_fused_plan(0, 256, None, None, None) == -1


def test_module_mirrors_reference_interface():
    from objectpermanence_amd import ModelsFactory, supported_models
    cfg = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}
    m = ModelsFactory.get_model("opnet", cfg)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    assert shapes == {
        "object_to_track_LSTM.weight_ih_l0": (1024, 90), "object_to_track_LSTM.weight_hh_l0": (1024, 256),
        "object_to_track_prediction.weight": (15, 256), "video_LSTM.weight_ih_l0": (2048, 6),
        "video_LSTM.weight_hh_l0": (2048, 512), "prediction_layer.weight": (4, 512)}
    assert sum(p.numel() for p in m.parameters()) == 1_421_056
    with pytest.raises(AttributeError, match="Model name is incorrect"):
        ModelsFactory.get_model("nope", cfg)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, 2, 15, 6))
    assert "opnet" in supported_models.DOUBLE_OUTPUT_MODELS


def _dp_worker(rank, world, port, n_total, tmp):
    import torch.distributed as dist
    from objectpermanence_amd import parallel
    from oracle import opnet_oracle as oo, synth
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    cfg = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 16, "videos_hidden_dim": 32}
    params = synth.opnet_synth_params(cfg)
    lo, hi = parallel.shard_range(n_total, world, rank)
    if hi > lo:
        boxes, _ = synth.make_batch(lo, hi - lo, 10)
        y, _ = oo.opnet_forward(boxes, params, np.float32)     # stand-in for the per-rank GPU forward
        local = torch.from_numpy(oo.postprocess_to_pixels(y))
    else:
        local = torch.zeros((0, 10, 4), dtype=torch.int32)
    full, _ = parallel.all_gather_predictions(local, n_total)
    np.save(os.path.join(tmp, f"r{rank}.npy"), full.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [5, 4, 1])
def test_dp_shard_and_gather_equals_single_process(tmp_path, n_total):
    import torch.multiprocessing as mp
    from oracle import opnet_oracle as oo, synth
    port = 29500 + (os.getpid() + n_total) % 1000
    mp.spawn(_dp_worker, args=(2, port, n_total, str(tmp_path)), nprocs=2, join=True)
    cfg = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 16, "videos_hidden_dim": 32}
    boxes, _ = synth.make_batch(0, n_total, 10)
    y, _ = oo.opnet_forward(boxes, synth.opnet_synth_params(cfg), np.float32)
    ref = oo.postprocess_to_pixels(y)
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), f"r{r}.npy"))
        assert np.array_equal(got, ref)


def test_shard_ranges_cover_everything():
    from objectpermanence_amd import parallel
    for n in (0, 1, 7, 8, 9, 255, 256):
        for w in (1, 2, 8):
            spans = [parallel.shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def _dp_grad_worker(rank, world, port, n_total, tmp):
    import torch.distributed as dist
    from objectpermanence_amd import parallel
    from oracle import synth, torch_port
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    cfg = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 16, "videos_hidden_dim": 32}
    params = synth.opnet_synth_params(cfg)
    lo, hi = parallel.shard_range(n_total, world, rank)
    boxes, labels = synth.make_batch(lo, hi - lo, 8)
    # stand-in for the per-rank GPU training step: local mean loss -> local gradients
    _, grads, _ = torch_port.loss_and_grads(boxes, labels, params)
    ps = [torch.nn.Parameter(torch.from_numpy(params[k].copy())) for k in params]
    for p_, k in zip(ps, params):
        p_.grad = torch.from_numpy(grads[k].copy())
    flat, _ = parallel.all_reduce_gradients(ps, hi - lo, n_total)
    parallel.unflatten_gradients(ps, flat)
    np.savez(os.path.join(tmp, f"g{rank}.npz"), **{k: p_.grad.numpy() for k, p_ in zip(params, ps)})
    dist.destroy_process_group()


def test_dp_gradient_allreduce_equals_single_process(tmp_path):
    """8x32 == 1x256 in miniature: 2 ranks with an UNEVEN split (2 + 1 clips) reproduce the gradient of the
    single-process mean loss over all 3 clips."""
    import torch.multiprocessing as mp
    from oracle import synth, torch_port
    port = 30500 + os.getpid() % 1000
    mp.spawn(_dp_grad_worker, args=(2, port, 3, str(tmp_path)), nprocs=2, join=True)
    cfg = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 16, "videos_hidden_dim": 32}
    boxes, labels = synth.make_batch(0, 3, 8)
    _, ref, _ = torch_port.loss_and_grads(boxes, labels, synth.opnet_synth_params(cfg))
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), f"g{r}.npz"))
        for k in ref:
            assert np.abs(got[k] - ref[k]).max() <= 1e-6 * max(1.0, np.abs(ref[k]).max()), k


def test_detector_score_filter_matches_reference(golden_dir):
    """detector.py:14-28 + the int cast of preprocess_perception_main.py:35 (row a11), vs goldens produced by the
    reference's own static method (tests/golden/detector_filter.json)."""
    import json
    from objectpermanence_amd.detector import CaterObjectDetector
    cases = json.load(open(os.path.join(golden_dir, "detector_filter.json")))
    assert len(cases) == 3
    for c in cases:
        out = CaterObjectDetector.remove_low_probability_object(
            {"boxes": torch.tensor(c["boxes"]), "labels": torch.tensor(c["labels"]), "scores": torch.tensor(c["scores"])})
        assert out["scores"].shape[0] == c["kept"] == out["labels"].shape[0]
        assert out["boxes"].numpy().astype(int).tolist() == c["kept_boxes_int"]


def test_batch_granular_sharding_for_transformer():
    from objectpermanence_amd import parallel
    for n, bs, w in ((37, 16, 2), (5, 16, 8), (64, 16, 4), (0, 4, 2)):
        owned = [parallel.shard_batches(n, bs, w, r) for r in range(w)]
        flat = sorted(x for part in owned for x in part)
        # every reference batch appears exactly once, intact, in dataset order
        assert flat == [(b * bs, min(n, (b + 1) * bs)) for b in range((n + bs - 1) // bs)]
        for r, part in enumerate(owned):
            assert all((lo // bs) % w == r for lo, _ in part)


def test_detector_factory_entry():
    from objectpermanence_amd import ModelsFactory
    det = ModelsFactory.get_detector_model("object_detector", "detection_model.pth")
    assert det.num_classes == 193 and det.saved_detector_path == "detection_model.pth"
    assert ModelsFactory.get_detector_model("nope") is None


def test_proj_utils_grid_classes():
    """reference proj_utils.py:37-75 without cv2: the homography is the closed-form inverse of the floor->image map"""
    from objectpermanence_amd import proj_utils as pu
    pts = np.array([[-3, -3, pu.Z], [0, 3, pu.Z], [-3, 0, pu.Z], [0, 0, pu.Z]])          # the reference's 4 fit points
    img = pu.project_3d_point(pts)
    q = pu.H @ np.vstack([img.T, np.ones(4)])
    assert np.abs((q[:2] / q[2]).T - pts[:, :2]).max() < 1e-9 and pu.H[2, 2] == 1.0
    centres = np.array([[gx - 2.5, gy - 2.5, pu.Z] for gy in range(6) for gx in range(6)])
    ic = pu.project_3d_point(centres)
    assert pu.get_class_predictions(ic[:, 0], ic[:, 1]).tolist() == list(range(36))        # cls = y1 * 6 + x1
    assert pu.get_class_prediction(float(ic[7, 0]), float(ic[7, 1])) == 7
    far = pu.project_3d_point(np.array([[40.0, -40.0, pu.Z]]))                             # clipped into the grid
    assert pu.get_class_prediction(far[0, 0], far[0, 1]) == 5


def test_cli_mirrors_reference_subcommands(tmp_path):
    """python -m objectpermanence_amd: the reference's five sub-commands with its flags (main.py:13-84); the analysis
    command runs end to end on CPU (the others need the GPU)"""
    from objectpermanence_amd.__main__ import build_parser, main
    p = build_parser()
    a = p.parse_args(["training", "--model_type", "opnet", "--model_config", "m.json", "--training_config", "t.json"])
    assert (a.mode, a.model_type, a.model_config, a.training_config) == ("training", "opnet", "m.json", "t.json")
    a = p.parse_args(["inference", "--model_type", "transformer_lstm", "--results_dir", "r", "--inference_config", "i.json"])
    assert a.model_config is None
    a = p.parse_args(["cater_inference", "--results_dir", "r", "--inference_config", "i.json", "--model_config", "m.json"])
    assert a.mode == "cater_inference"
    assert p.parse_args(["preprocess", "--results_dir", "r", "--config", "c.json"]).config == "c.json"
    with pytest.raises(SystemExit):
        p.parse_args(["inference", "--model_type", "DaSiamRPN", "--results_dir", "r", "--inference_config", "i.json"])
    pred, lab = tmp_path / "pred", tmp_path / "lab"
    pred.mkdir(); lab.mkdir()
    rng = np.random.default_rng(0)
    for v in ("v0", "v1"):
        gt = np.tile(np.array([[50, 60, 30, 40]]), (300, 1))                       # xywh labels
        json.dump({"small_gold_spl_metal_Spl_0": gt.tolist()}, open(lab / f"{v}_bb.json", "w"))
        box = np.tile(np.array([[50, 60, 80, 100]]), (300, 1)) + rng.integers(-3, 4, size=(300, 4))
        json.dump(box.tolist(), open(pred / f"{v}_bb.json", "w"))
    out = tmp_path / "res.csv"
    assert main(["analysis", "--predictions_dir", str(pred), "--labels_dir", str(lab), "--iou_thresholds", "0.5,0.9",
                 "--output_file", str(out)]) == 0
    rows = open(out).read().strip().splitlines()
    assert len(rows) >= 3 and "overall" in rows[0]


def test_bench_gpus_flag_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must start 2 ranks itself (torch.distributed.run, rendezvous on
    127.0.0.1); --launcher-selftest stops after the rendezvous + one gloo collective, so this runs without a GPU."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--launcher-selftest"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2


def test_bench_train_mode_launches_ranks_and_reduces_the_gradient_bucket():
    """config 5's line without the hardware: `bench.py --gpus 2 --mode train` starts its own ranks; the self-test leg runs the
    exchange the real run does (the flat 1 421 056-float OPNet gradient bucket + its guard slot through
    parallel.GradBucket.all_reduce) on gloo and reports global batch / parallelism as the real line would."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--mode", "train", "--launcher-selftest"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["mode"] == "train" and out["grad_bucket_floats"] == 1_421_056 and out["guard_after_allreduce"] == 1.0
    assert out["config"] == {"global_batch": 64, "parallelism": "dp2"} and out["scaling"] == "weak"
    # config 5 at its fixed global batch: 256 clips over the ranks, "strong"
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--mode", "train", "--global-batch", "256",
                        "--launcher-selftest"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["config"] == {"global_batch": 256, "parallelism": "dp2"} and out["scaling"] == "strong" and out["batch_per_gpu"] == 128


def test_bench_refuses_more_gpus_than_the_node_has():
    import subprocess
    import sys
    import torch
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", str(have + 1 if have else 2)],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and "refusing" in (r.stderr + r.stdout)
    assert not any(l.startswith("{") for l in r.stdout.splitlines())


def test_reasoner_server_batches_requests_and_slices_outputs():
    """request batching (serving.ReasonerServer) with a stand-in model on CPU: several submitted minibatches run as ONE
    forward and every request gets exactly its own rows back"""
    import torch
    from objectpermanence_amd.serving import ReasonerServer

    calls = []

    class Fake(torch.nn.Module):
        def forward(self, x):
            calls.append(int(x.shape[0]))
            return x[:, :, 0, :4] * 2.0, x[:, :, :, 0].permute(0, 2, 1)

    srv = ReasonerServer(Fake(), "opnet", max_clips=96)
    xs = [torch.randn(n, 5, 15, 6) for n in (32, 16, 48, 8)]
    hs = [srv.submit(x) for x in xs[:3]]            # 32 + 16 + 48 = 96 -> flushes itself
    assert calls == [96] and all(h.done() for h in hs)
    h4 = srv.submit(xs[3])
    assert not h4.done()
    y4, lg4 = h4.result()                           # flushes the remainder
    assert calls == [96, 8] and srv.forwards == 2 and srv.clips == 104
    for x, h in zip(xs, hs + [h4]):
        y, lg = h.result()
        assert torch.equal(y, x[:, :, 0, :4] * 2.0) and torch.equal(lg, x[:, :, :, 0].permute(0, 2, 1))
    # a request with another clip length cannot share the launch
    a, b = srv.submit(torch.randn(4, 5, 15, 6)), srv.submit(torch.randn(4, 7, 15, 6))
    assert a.done() and not b.done()
    srv.flush()
    assert b.done() and calls[-2:] == [4, 4]
    # before_launch (a data-parallel caller makes the launch wait for its previous collective there): once per forward, after the
    # requests were concatenated and before the model is called
    order = []
    srv.before_launch = lambda: order.append(("hook", len(calls)))
    srv.submit(torch.randn(8, 5, 15, 6)); srv.submit(torch.randn(8, 5, 15, 6))
    srv.flush()
    assert order == [("hook", len(calls) - 1)] and calls[-1] == 16


# ---------------------------------------------------------------------------------------------------------------------
# data-parallel drivers (round 2): every rank always joins, balanced slices, whole batches for transformer_lstm
# ---------------------------------------------------------------------------------------------------------------------
def test_balanced_range_leaves_no_rank_empty():
    from objectpermanence_amd import parallel
    for n in (0, 1, 4, 5, 7, 16, 33):
        for w in (1, 2, 4, 8):
            spans = [parallel.balanced_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n and all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
            if n >= w:
                assert min(sizes) >= 1          # shard_range(5, 4, 3) == (5, 5): the case that used to hang training


def test_training_batches_plan():
    from objectpermanence_amd.training_main import training_batches
    # clip-independent model: every step is one reference batch, split; the last batch of 5 on 4 ranks leaves nobody empty
    plans = [training_batches("opnet", 21, 16, 4, r) for r in range(4)]
    assert all(len(p) == 2 for p in plans) and [p[1][1] for p in plans] == [5] * 4
    for k in range(2):
        got = sorted(i for p in plans for i in p[k][0])
        assert got == list(range(16 * k, min(21, 16 * k + 16)))
    assert all(len(p[1][0]) >= 1 for p in plans)
    # fewer clips than ranks: some ranks get nothing for that step but still have the step (and join its collective)
    plans = [training_batches("opnet", 2, 16, 4, r) for r in range(4)]
    assert [len(p) for p in plans] == [1] * 4 and sorted(len(p[0][0]) for p in plans) == [0, 0, 1, 1]
    # transformer_lstm: whole reference batches, one optimiser step spans `world` of them
    plans = [training_batches("transformer_lstm", 40, 16, 2, r) for r in range(2)]
    assert [p[0][0] for p in plans] == [list(range(0, 16)), list(range(16, 32))] and plans[0][0][1] == 32
    assert plans[0][1] == (list(range(32, 40)), 8) and plans[1][1] == ([], 8)


def _dp_train_worker(rank, world, port, n_total, tmp):
    """training.train_step on CPU tensors over gloo with a stand-in model (the OPNet graph on torch ops): rank 1's slice of
    the batch is EMPTY when n_total == 1 - it must still join the all-reduce and take the same Adam step"""
    import torch.distributed as dist
    from objectpermanence_amd import parallel, training
    from oracle import synth, torch_port
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    cfg = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 16, "videos_hidden_dim": 32}
    model = torch_port.OPNetTorch(synth.opnet_synth_params(cfg))
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    lo, hi = parallel.balanced_range(n_total, world, rank)
    boxes, labels = synth.make_batch(0, n_total, 8)
    old = training.compute_loss
    training.compute_loss = lambda name, out, lab, mask=None, kind="l1", **kw: (torch.mean(torch.abs(out - lab)),) * 3
    try:
        xb = torch.from_numpy(boxes[lo:hi]) if hi > lo else None
        lb = torch.from_numpy(labels[lo:hi]) if hi > lo else None
        training.train_step("baseline_lstm", model, opt, xb, lb, n_global=n_total)   # single-output stand-in
    finally:
        training.compute_loss = old
    np.savez(os.path.join(tmp, f"w{rank}.npz"), **{k: v.detach().numpy() for k, v in model.state_dict().items()})
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [3, 1])
def test_dp_train_step_joins_with_empty_shard_and_matches_single_process(tmp_path, n_total):
    import torch.multiprocessing as mp
    from oracle import synth, torch_port
    port = 31500 + (os.getpid() + n_total) % 1000
    mp.spawn(_dp_train_worker, args=(2, port, n_total, str(tmp_path)), nprocs=2, join=True)   # would hang before the fix
    cfg = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 16, "videos_hidden_dim": 32}
    model = torch_port.OPNetTorch(synth.opnet_synth_params(cfg))
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    boxes, labels = synth.make_batch(0, n_total, 8)
    opt.zero_grad()
    torch.mean(torch.abs(model(torch.from_numpy(boxes)) - torch.from_numpy(labels))).backward()
    opt.step()
    ref = {k: v.detach().numpy() for k, v in model.state_dict().items()}
    for r in range(2):
        got = np.load(os.path.join(str(tmp_path), f"w{r}.npz"))
        for k in ref:
            assert np.abs(got[k] - ref[k]).max() <= 2e-6, (r, k)      # both ranks hold the single-process weights


def _dp_transformer_worker(rank, world, port, n_total, bs, tmp):
    import torch.distributed as dist
    from objectpermanence_amd import parallel
    from oracle import synth, torch_port
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    cfg = {"boxes_features_dim": 16, "num_attention_heads": 2, "num_attention_layers": 1, "num_lstm_layers": 1, "lstm_hidden_dim": 16}
    params = {k: torch.from_numpy(v) for k, v in synth.transformer_lstm_synth_params(cfg, ffn=64).items()}
    x_all, _ = synth.make_batch(0, n_total, 6)
    x_all = torch.from_numpy(synth.boxes5(x_all))
    for tag, plan in (("whole", parallel.plan_inference_batches("transformer_lstm", n_total, bs, world, rank)),
                      ("split", parallel.plan_inference_batches("opnet", n_total, bs, world, rank))):
        rows, index = [], []
        for b in plan:
            with torch.no_grad():
                rows.append(torch_port.transformer_lstm_forward(x_all[b], params, cfg["num_attention_heads"]))
            index.extend(b)
        local = torch.cat(rows) if rows else torch.zeros((0, 6, 4))
        full = parallel.all_gather_by_index(local, torch.tensor(index, dtype=torch.int64), n_total)
        np.save(os.path.join(tmp, f"{tag}{rank}.npy"), full.numpy())
    dist.destroy_process_group()


def test_dp_transformer_lstm_needs_whole_reference_batches(tmp_path):
    """TransformerLstm attends across the clips of a minibatch (reference learned_models.py:183-185): two ranks that take
    WHOLE reference batches reproduce the single-process outputs; the contiguous clip split that is fine for OPNet cuts a
    batch in two (5 clips, batch 4, 2 ranks: rank 0 gets clips 0-2) and changes them."""
    import torch.multiprocessing as mp
    from oracle import synth, torch_port
    n_total, bs = 5, 4
    port = 32500 + os.getpid() % 1000
    mp.spawn(_dp_transformer_worker, args=(2, port, n_total, bs, str(tmp_path)), nprocs=2, join=True)
    cfg = {"boxes_features_dim": 16, "num_attention_heads": 2, "num_attention_layers": 1, "num_lstm_layers": 1, "lstm_hidden_dim": 16}
    params = {k: torch.from_numpy(v) for k, v in synth.transformer_lstm_synth_params(cfg, ffn=64).items()}
    x_all, _ = synth.make_batch(0, n_total, 6)
    x_all = torch.from_numpy(synth.boxes5(x_all))
    with torch.no_grad():
        ref = torch.cat([torch_port.transformer_lstm_forward(x_all[b:b + bs], params, 2) for b in range(0, n_total, bs)]).numpy()
    for r in range(2):
        assert np.abs(np.load(os.path.join(str(tmp_path), f"whole{r}.npy")) - ref).max() < 1e-6
        assert np.abs(np.load(os.path.join(str(tmp_path), f"split{r}.npy")) - ref).max() > 1e-4


def test_masked_mean_iou_skips_videos_without_containment_frames():
    """np.mean over the reference's DataFrame column is Series.mean(skipna=True) (training_main.py:105-112)"""
    import torch
    from objectpermanence_amd.training_main import masked_mean_iou
    iou = torch.tensor([[0.5, 1.0, 0.0], [0.2, 0.4, 0.6], [0.9, 0.9, 0.9]], dtype=torch.float64)
    cm = torch.tensor([[True, True, False], [False, False, False], [False, False, True]])
    assert abs(masked_mean_iou(iou, cm) - (0.75 + 0.9) / 2) < 1e-12
    assert np.isnan(masked_mean_iou(iou, torch.zeros_like(cm)))


def test_grid_classes_match_reference_proj_utils(golden_dir):
    """6x6-grid snitch localisation (reference baselines/proj_utils.py:37-75, run as written under a numpy-DLT stand-in for
    cv2.findHomography - oracle/homography.py, oracle/gen_golden.py): the closed-form homography of
    objectpermanence_amd/proj_utils.py gives the same H and the same class for every lattice / projected floor point that
    is not within rounding of a cell border."""
    from objectpermanence_amd import proj_utils as pu
    from oracle import homography
    g = np.load(os.path.join(golden_dir, "grid_classes.npz"))
    assert np.abs(pu.H - g["H"]).max() < 1e-9
    cls = pu.get_class_predictions(g["cx"], g["cy"])
    q = g["H"] @ np.stack([g["cx"], g["cy"], np.ones_like(g["cx"])])
    xy = q[:2] / q[2]
    # floor() is discontinuous at the cell borders: skip points within rounding of one (points beyond the grid are clipped
    # to exactly -3 / 3 - 1e-5 by both and are safe)
    away = np.all((np.abs(xy - np.round(xy)) > 1e-6) | (xy < -3.001) | (xy > 3.001), axis=0)
    assert away.sum() > 2000 and np.array_equal(cls[away], g["cls"][away])
    # the stand-in itself: four exact correspondences are reproduced
    pts3 = np.array([[-3, -3, pu.Z], [0, 3, pu.Z], [-3, 0, pu.Z], [0, 0, pu.Z]])
    Hd, _ = homography.find_homography_dlt(pu.project_3d_point(pts3), pts3[:, :2])
    back = homography.perspective_transform(pu.project_3d_point(pts3).reshape(-1, 1, 2), Hd).reshape(-1, 2)
    assert np.abs(back - pts3[:, :2]).max() < 1e-9


def test_small_batch_persistent_form_sizes_and_limits():
    """host-only entry points of the 4-clip persistent kernels (no GPU needed): workspace / packed sizes, the batch limit, the
    refusal of other hidden sizes"""
    from objectpermanence_amd import _lib
    lib = _lib.load()
    assert lib.opnet_xcd4_max_batch() == 128
    assert lib.opnet_xcd4_packed_weights_bytes(256, 512) == (2048 * 512 + 1024 * 352 + 16 * 256 + 2048 * 8 + 32 * 256) * 4
    assert lib.opnet_xcd4_packed_weights_bytes(128, 512) == 0
    w16, w32, w33 = (lib.opnet_xcd4_workspace_bytes(b, 300, 256, 512) for b in (16, 32, 33))
    assert 0 < w16 == w32 < w33                       # whole row blocks of 32 clips
    assert lib.opnet_xcd4_workspace_bytes(129, 300, 256, 512) == 0
    assert lib.opnet_xcd4_workspace_bytes(32, 300, 256, 256) == 0
    # the training workspace carries the exchange rings of the persistent step for batches it can serve
    t32, t256 = lib.opnet_train_workspace_bytes(32, 300, 256, 512), lib.opnet_train_workspace_bytes(256, 300, 256, 512)
    fixed = 1024 * 16384 * 4                              # partial tiles of the weight-gradient waves: one round of the SIMDs, any batch
    assert t32 - fixed > 32 * 5_000_000 and t256 - fixed > 7 * (t32 - fixed) * 0.9


# ---- data parallelism through the product entry points (parallel.init_from_env) --------------------

_ENTRY_RANK = r"""
import json, os, sys
import torch, torch.distributed as dist
from objectpermanence_amd import parallel
with parallel.init_from_env() as launch:
    w, r, active = parallel.world_rank()
    t = torch.tensor([float(r + 1)])
    dist.all_reduce(t)
    out = {"owned": launch.owned, "backend": dist.get_backend(), "world": w, "rank": r, "active": active, "sum": float(t),
           "device": str(parallel.resolve_device("cpu"))}
    open(os.path.join(sys.argv[1], f"rank{r}.json"), "w").write(json.dumps(out))
assert not dist.is_initialized()
"""


def test_entry_points_join_the_job_torchrun_describes(tmp_path):
    """WORLD_SIZE / RANK / LOCAL_RANK in the environment -> the process joins the group (gloo here: no GPU), every collective
    spans the ranks, the group is left again at exit; the JSON's device stands when there is no GPU to bind"""
    import subprocess
    import sys
    script = tmp_path / "rank.py"
    script.write_text(_ENTRY_RANK)
    port = str(33500 + os.getpid() % 1000)
    procs = []
    for r in range(2):
        env = dict(os.environ, WORLD_SIZE="2", RANK=str(r), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=port,
                   PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        env.pop("OPNET_FORCE_DIST", None)
        procs.append(subprocess.Popen([sys.executable, str(script), str(tmp_path)], env=env, stderr=subprocess.PIPE, text=True))
    for p in procs:
        _, err = p.communicate(timeout=180)
        assert p.returncode == 0, err[-2000:]
    import json
    for r in range(2):
        out = json.load(open(tmp_path / f"rank{r}.json"))
        assert out == {"owned": True, "backend": "gloo", "world": 2, "rank": r, "active": True, "sum": 3.0, "device": "cpu"}


def test_init_from_env_is_a_no_op_outside_a_job(monkeypatch):
    import torch.distributed as dist
    from objectpermanence_amd import parallel
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "OPNET_FORCE_DIST"):
        monkeypatch.delenv(k, raising=False)
    with parallel.init_from_env() as launch:
        assert not launch.owned and not dist.is_initialized()
        assert parallel.world_rank() == (1, 0, False) and not parallel.is_active()
        assert parallel.resolve_device("cuda:3") == torch.device("cuda:3")     # the JSON's device, as the reference reads it


def _forced_world_of_one_worker(rank, world, port, tmp):
    """OPNET_FORCE_DIST=1 in a gloo group of ONE rank: train_step takes the data-parallel branch (bucket all-reduce, weight
    n_local / n_global) and leaves the weights of the plain step"""
    import torch.distributed as dist
    from objectpermanence_amd import parallel, training
    from oracle import synth, torch_port
    os.environ.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
    cfg = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 16, "videos_hidden_dim": 32}
    boxes, labels = (torch.from_numpy(a) for a in synth.make_batch(0, 3, 8))
    training.compute_loss = lambda name, out, lab, mask=None, kind="l1", **kw: (torch.mean(torch.abs(out - lab)),) * 3
    res = {}
    for forced in ("0", "1"):
        os.environ["OPNET_FORCE_DIST"] = forced
        with parallel.init_from_env() as launch:
            # (WORLD_SIZE=1 alone joins a group of one too - torchrun with one rank - but its exchange stays off)
            assert launch.owned and dist.get_world_size() == 1 and parallel.is_active() == (forced == "1")
            model = torch_port.OPNetTorch(synth.opnet_synth_params(cfg))
            opt = torch.optim.Adam(model.parameters(), lr=1e-3)
            calls = []
            if forced == "1":
                real = dist.all_reduce
                dist.all_reduce = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
            for _ in range(2):
                training.train_step("baseline_lstm", model, opt, boxes, labels, n_global=3)
            if forced == "1":
                dist.all_reduce = real
                assert len(calls) == 2
            res[forced] = {k: v.detach().clone() for k, v in model.state_dict().items()}
    for k in res["0"]:
        assert torch.equal(res["0"][k], res["1"][k]), k
    open(os.path.join(tmp, "ok"), "w").write("ok")


def test_forced_group_of_one_takes_the_data_parallel_branch(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_forced_world_of_one_worker, args=(1, 34500 + os.getpid() % 1000, str(tmp_path)), nprocs=1, join=True)
    assert os.path.exists(tmp_path / "ok")


def _failing_rank_worker(rank, world, port, tmp):
    """rank 1 raises inside the launch context while rank 0 is in a collective: rank 1 must leave at once (no barrier)"""
    import time
    import torch.distributed as dist
    from objectpermanence_amd import parallel
    os.environ.update({"WORLD_SIZE": str(world), "RANK": str(rank), "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(port), "OPNET_DIST_BACKEND": "gloo"})
    t0 = time.time()
    try:
        with parallel.init_from_env():
            dist.all_reduce(torch.ones(1))                 # both ranks: the job is up
            if rank == 1:
                raise ValueError("a refused clip file")
            dist.all_reduce(torch.ones(4))                 # rank 0: a collective rank 1 never joins
    except ValueError:
        open(os.path.join(tmp, "failed_rank_left_after"), "w").write(f"{time.time() - t0:.1f}")
        raise SystemExit(3)
    except Exception:                                      # rank 0: its peer died under the collective - it must not hang either
        open(os.path.join(tmp, "peer_saw_the_failure"), "w").write("ok")
        raise SystemExit(4)


def test_a_failing_rank_leaves_without_waiting_at_a_barrier(tmp_path):
    """Launch.__exit__ used to call shutdown() - and with it dist.barrier() - while an exception was
    propagating; the failing rank then waited for peers that were inside other collectives until the watchdog fired.  Now the
    barrier is on the clean path only: the failing rank is out within seconds with its own exception."""
    import time
    import torch.multiprocessing as mp
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    t0 = time.time()
    ctx = mp.spawn(_failing_rank_worker, args=(2, port, str(tmp_path)), nprocs=2, join=False)
    procs = ctx.processes
    procs[1].join(60)
    assert not procs[1].is_alive(), "the failing rank is still waiting (barrier under an exception?)"
    assert procs[1].exitcode == 3
    assert float(open(tmp_path / "failed_rank_left_after").read()) < 30 and time.time() - t0 < 60
    procs[0].join(60)                                      # gloo notices the closed connection
    if procs[0].is_alive():
        procs[0].terminate()
        procs[0].join()


def test_launch_monitor_reaps_completed_entries_and_recycles_slots():
    """a watch entry (and the input its `redo` holds) goes as soon as its launch is seen complete and clean;
    an aborted one stays for verify(); slots come from a free list"""
    from objectpermanence_amd import launch_monitor as lm

    class Ev:
        def __init__(self, done):
            self.done = done

        def query(self):
            return self.done

        def synchronize(self):
            self.done = True

    mon = lm.LaunchMonitor()
    mon._host = torch.zeros((lm._SLOTS, 4), dtype=torch.int32)
    held = []
    for i, (done, code) in enumerate([(True, 0), (False, 0), (True, 3), (True, 0)]):
        slot = mon._free.pop()
        mon._host[slot, 0] = code
        mon._pending.append((Ev(done), slot, (lambda i=i: held.append(i)), f"launch {i}"))
    assert mon.reap() == 2 and mon.pending() == 2 and len(mon._free) == lm._SLOTS - 2
    import warnings
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        assert mon.verify() == 1
    assert held == [2] and mon.pending() == 0 and sorted(mon._free) == list(range(lm._SLOTS))


def test_server_merges_segmenting_models_by_request_not_by_concatenation():
    """host logic of ReasonerServer for a model whose clips are coupled inside a request (TransformerLstm): pending requests of one
    shape go to forward_segments(x, n) - never to a plain forward of the concatenation -, a request of another shape or the
    model's per-pass limit cuts the pass, a lone request takes the plain forward (stub model, no GPU)"""
    from objectpermanence_amd.serving import ReasonerServer

    class Stub(torch.nn.Module):
        calls = []

        def max_requests_per_pass(self, b, T, exact=False):
            return 3

        def forward(self, x):
            Stub.calls.append(("plain", tuple(x.shape)))
            return x.sum(dim=(2, 3), keepdim=False).unsqueeze(-1).repeat(1, 1, 4)

        def forward_segments(self, x, n, exact=False):
            Stub.calls.append(("segments", tuple(x.shape), n))
            return self.forward(x)

    m = Stub()
    server = ReasonerServer(m, "transformer_lstm")
    assert server.segmented
    reqs = [torch.full((1, 5, 15, 5), float(i)) for i in range(4)] + [torch.full((2, 5, 15, 5), 9.0)]
    hs = [server.submit(r) for r in reqs]           # the third submit fills a pass (limit 3); the fifth has another shape
    outs = [h.result() for h in hs]
    kinds = [c for c in Stub.calls if c[0] == "segments"]
    assert kinds == [("segments", (3, 5, 15, 5), 3)]                       # one merged pass of three requests
    assert ("plain", (1, 5, 15, 5)) in Stub.calls and ("plain", (2, 5, 15, 5)) in Stub.calls   # the lone ones: plain forwards
    for r, o in zip(reqs, outs):
        assert o.shape == (r.shape[0], 5, 4) and float(o[0, 0, 0]) == float(r[0, 0].sum())
    assert server.forwards == 3


def _broadcast_worker(rank, world, port, tmp):
    import torch.distributed as dist
    from objectpermanence_amd import parallel
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                       # N processes of the reference's unseeded construction: N different models
    m = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.BatchNorm1d(3))
    m[1].running_mean.fill_(float(rank + 1))
    before = [p.detach().clone() for p in m.parameters()]
    parallel.broadcast_parameters(m)
    torch.save({"after": [t.detach().clone() for t in list(m.parameters()) + list(m.buffers())], "before": before},
               os.path.join(tmp, f"b{rank}.pt"))
    dist.destroy_process_group()


def test_every_rank_starts_from_rank_zeros_weights(tmp_path):
    """training_main builds its model with torch's random initialisation on every rank: broadcast_parameters makes them one model
    (without it, averaging the gradients of N different models trains none of them); a no-op outside a job"""
    import torch.multiprocessing as mp
    from objectpermanence_amd import parallel
    mp.spawn(_broadcast_worker, args=(2, 36100 + os.getpid() % 1000, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(os.path.join(str(tmp_path), f"b{r}.pt")) for r in range(2))
    assert not torch.equal(r0["before"][0], r1["before"][0])
    assert all(torch.equal(a, b) for a, b in zip(r0["after"], r1["after"]))                 # parameters AND buffers
    assert all(torch.equal(a, b) for a, b in zip(r0["after"][:len(r0["before"])], r0["before"]))   # ... rank 0's
    lone = torch.nn.Linear(2, 2)
    w = lone.weight.detach().clone()
    parallel.broadcast_parameters(lone)
    assert torch.equal(lone.weight, w)


def test_cv2_branch_reads_frame_count_minus_one_frames(tmp_path, monkeypatch):
    """reference tracking_utils.VideoHandling (:27-30, :44-45): cv2 "always returns an extra frame", the labels align with the first
    300, so CAP_PROP_FRAME_COUNT - 1 frames are read.  cv2 is absent from this image: a stub module stands in for it."""
    import sys
    import types
    from objectpermanence_amd.preprocess_perception_main import read_video_frames

    class Cap:
        released = 0

        def __init__(self, path, n):
            self.n, self.i = n, 0

        def isOpened(self):
            return self.n >= 0

        def get(self, prop):
            assert prop == 7
            return float(self.n)

        def read(self):
            if self.i >= self.n:
                return False, None
            self.i += 1
            return True, np.full((2, 2, 3), (self.i - 1) % 256, dtype=np.uint8)

        def release(self):
            Cap.released += 1

    counts = {"a.avi": 301, "short.avi": 5, "empty.avi": 0, "bad.avi": -1}
    cv2 = types.ModuleType("cv2")
    cv2.CAP_PROP_FRAME_COUNT = 7
    cv2.VideoCapture = lambda path: Cap(path, counts[os.path.basename(path)])
    monkeypatch.setitem(sys.modules, "cv2", cv2)
    frames = list(read_video_frames(tmp_path / "a.avi"))
    assert len(frames) == 300 and int(frames[0][0, 0, 0]) == 0 and int(frames[-1][0, 0, 0]) == 299 % 256
    assert len(list(read_video_frames(str(tmp_path / "short.avi")))) == 4
    assert list(read_video_frames(tmp_path / "empty.avi")) == []
    assert Cap.released == 3
    with pytest.raises(RuntimeError, match="Unable to open"):
        list(read_video_frames(tmp_path / "bad.avi"))


def test_deferred_consumer_consumes_in_order_when_ready_and_bounds_what_is_alive():
    """launch_monitor.DeferredConsumer (the evaluation / inference loops): an output is post-processed and dropped as soon as its
    event has completed - never before, never out of order - every completed launch is settled first (an aborted one healed
    before its output is used), and more than max_pending waiting entries make the producer wait for the oldest"""
    from objectpermanence_amd import launch_monitor as lm

    class Ev:
        def __init__(self, done=False):
            self.done, self.waited = done, 0

        def query(self):
            return self.done

        def synchronize(self):
            self.waited += 1
            self.done = True

    class Model(torch.nn.Module):
        pass

    m = Model()
    m._monitor = lm.LaunchMonitor()
    m._monitor._host = torch.zeros((lm._SLOTS, 4), dtype=torch.int32)
    healed, got = [], []
    # a watched launch that aborted and has completed: it must be healed before the first consume
    slot = m._monitor._free.pop()
    m._monitor._host[slot, 0] = 1
    m._monitor._pending.append((Ev(True), slot, lambda: healed.append("redo"), "launch"))
    running = Ev(False)                               # ... and one still running: settle() must not wait for it
    slot2 = m._monitor._free.pop()
    m._monitor._pending.append((running, slot2, None, "running"))
    d = lm.DeferredConsumer(m, lambda tag: got.append((tag, list(healed))), max_pending=3)
    evs = [Ev(False) for _ in range(6)]
    import warnings
    with warnings.catch_warnings(record=True):
        warnings.simplefilter("always")
        d.add(evs[0], "a")
        d.add(evs[1], "b")
        assert got == []                                  # nothing is ready: nothing consumed, nobody waited
        evs[1].done = True
        d.drain()
        assert got == []                                  # b is ready but a is not: order is kept
        evs[0].done = True
        d.drain()
    assert got == [("a", ["redo"]), ("b", ["redo"])] and running.waited == 0 and m._monitor.pending() == 1
    for k in range(2, 6):
        d.add(evs[k], "cdef"[k - 2])                      # the fourth waiting entry exceeds max_pending = 3: waits for the oldest
    assert evs[2].waited == 1 and [g[0] for g in got] == ["a", "b", "c"] and d.peak_pending == 4
    d.drain(block=True, all_=True)
    assert [g[0] for g in got] == list("abcdef") and all(e.done for e in evs)
