"""BASELINE config 5's DEVICE path on the one GPU a test box has: an RCCL process group of ONE rank with OPNET_FORCE_DIST=1 takes
every data-parallel branch of the product - comm stream, in-place gradient bucket, RCCL all-reduce, guard slots, guarded
Adam through `guard_ptr`, gathers by dataset index, rank-0 writes - and must leave exactly what the plain single-process run
leaves (a sum over one rank and a weight of n/n are identities, so "exactly" is bit-for-bit).  The reference has no
distributed path (training_main.py:144,162: one device from the JSON); its contract for a sharded run is N-GPU == 1-GPU.

The entry-point tests start `python -m torch.distributed.run ... -m objectpermanence_amd ...` the way INTEGRATION.md documents,
with a JSON `device` that does not exist on the box: the rank must run on cuda:LOCAL_RANK instead."""
import json
import os
import pickle
import subprocess
import sys
import warnings

import numpy as np
import pytest
import torch

from oracle import opnet_oracle as oo, synth

pytestmark = pytest.mark.gpu
CFG = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def rccl_world_of_one():
    """an RCCL group of one rank on cuda:0, joined the way the product joins it (parallel.init_from_env)"""
    import torch.distributed as dist
    from objectpermanence_amd import parallel
    assert not dist.is_initialized()
    saved = {k: os.environ.get(k) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", parallel.FORCE_ENV)}
    os.environ.update({"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1",
                       "MASTER_PORT": str(29600 + os.getpid() % 300)})
    launch = parallel.init_from_env()
    assert launch.owned and dist.get_backend() == "nccl" and dist.get_world_size() == 1
    try:
        yield launch
    finally:
        parallel.shutdown(launch)
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        assert not dist.is_initialized()


def _model(train=True):
    from objectpermanence_amd import ModelsFactory
    params = synth.opnet_synth_params(CFG)
    m = ModelsFactory.get_model("opnet", CFG)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    m = m.to("cuda:0")
    return m.train(True) if train else m.eval()


def _three_steps(B, T, forced, monkeypatch, comm=None, events=None):
    from objectpermanence_amd import FusedAdam, parallel
    from objectpermanence_amd.training import global_loss, train_step
    if forced:
        monkeypatch.setenv(parallel.FORCE_ENV, "1")
    else:
        monkeypatch.delenv(parallel.FORCE_ENV, raising=False)
    assert parallel.is_active() == forced
    m = _model()
    opt = FusedAdam(m.parameters(), lr=1e-3)
    losses = []
    for k in range(3):
        boxes_np, labels_np = synth.make_batch(10 * k, B, T)
        loss = train_step("opnet", m, opt, torch.from_numpy(boxes_np).to("cuda:0"), torch.from_numpy(labels_np).to("cuda:0"),
                          n_global=B, comm_stream=comm, comm_events=events)
        losses.append((float(loss), float(global_loss(m, loss))))
    torch.cuda.synchronize()
    return m, opt, losses


@pytest.mark.parametrize("B,T", [(8, 10), (32, 300), (40, 7)])
def test_three_steps_through_the_rccl_branch_are_bit_identical(rccl_world_of_one, monkeypatch, B, T):
    """(32, 300) = the persistent training step of BASELINE configs 2 / 5; (40, 7) = the launch chain (above 32 clips)"""
    plain, popt, plosses = _three_steps(B, T, False, monkeypatch)
    comm, events = torch.cuda.Stream(device="cuda:0"), []
    dp, dopt, dlosses = _three_steps(B, T, True, monkeypatch, comm, events)
    assert len(events) == 3 and all(a.elapsed_time(b) > 0 for a, b in events)      # the collective ran, on the comm stream
    assert dp._grad_bucket.flat.data_ptr() == dp._grad_bucket._buf.data_ptr()
    for (n, a), (_, b) in zip(plain.named_parameters(), dp.named_parameters()):
        assert torch.equal(a.detach(), b.detach()), n
        assert b.grad.data_ptr() >= dp._grad_bucket.flat.data_ptr()                 # the gradients live in the bucket
    for a, b in zip(popt.state.values(), dopt.state.values()):
        assert int(a["step"]) == int(b["step"]) == 3
        assert torch.equal(a["exp_avg"], b["exp_avg"]) and torch.equal(a["exp_avg_sq"], b["exp_avg_sq"])
    for (lp, _), (ld, lg) in zip(plosses, dlosses):
        assert lp == ld == lg                 # guard slot 2: the all-reduced loss of the whole minibatch = the loss, at one rank
    g = dp._grad_bucket.guard.cpu().numpy()
    assert g[0] == 0 and g[1] == 0 and g[3] == 0
    assert dopt.abort_ptr is None and dopt.loss_ptr is None and dopt.guard_ptr is None      # cleared behind the step


def test_forced_abort_reaches_the_optimiser_through_the_guard_slot(rccl_world_of_one, monkeypatch):
    """OPNET_X4_DEBUG=4: the persistent forward gives up, its abort word travels guard slot 0 -> all-reduce -> guarded Adam
    (abort_ptr itself is NOT handed to the optimiser in data parallel), the weights stay, step_aborted() says so and the
    repeated step equals a clean one"""
    from objectpermanence_amd import FusedAdam, _lib, parallel
    from objectpermanence_amd.training import step_aborted, step_skipped_nonfinite, train_step
    lib = _lib.load()
    boxes_np, labels_np = synth.make_batch(3, 8, 10)
    boxes, labels = torch.from_numpy(boxes_np).to("cuda:0"), torch.from_numpy(labels_np).to("cuda:0")
    comm = torch.cuda.Stream(device="cuda:0")
    monkeypatch.setenv(parallel.FORCE_ENV, "1")
    try:
        lib.opnet_xcd4_enable(0)
        m0 = _model()
        o0 = FusedAdam(m0.parameters(), lr=1e-3)
        train_step("opnet", m0, o0, boxes, labels, n_global=8, comm_stream=comm)
        want = [p.detach().clone() for p in m0.parameters()]
        lib.opnet_xcd4_enable(1)

        m = _model()
        opt = FusedAdam(m.parameters(), lr=1e-3)
        before = [p.detach().clone() for p in m.parameters()]
        monkeypatch.setenv("OPNET_X4_DEBUG", "4")
        seen = {}
        real_step = opt.step

        def spy():
            seen.update(abort=opt.abort_ptr, loss=opt.loss_ptr, guard=opt.guard_ptr)
            return real_step()

        opt.step = spy
        loss = train_step("opnet", m, opt, boxes, labels, n_global=8, comm_stream=comm)
        torch.cuda.synchronize()
        monkeypatch.delenv("OPNET_X4_DEBUG")
        opt.step = real_step
        assert seen["abort"] is None and seen["loss"] is None and seen["guard"] == m._grad_bucket.guard.data_ptr()
        assert not np.isfinite(float(loss))
        g = m._grad_bucket.guard.cpu().numpy()
        assert g[0] == 1.0 and g[1] == 1.0              # the launch gave up AND (its output being NaN) the loss is not finite
        for p, b in zip(m.parameters(), before):
            assert torch.equal(p.detach(), b)
        with warnings.catch_warnings(record=True):
            warnings.simplefilter("always")
            assert step_aborted(m)
        assert lib.opnet_xcd4_enabled() == 0
        opt.rollback_step_count()
        loss2 = train_step("opnet", m, opt, boxes, labels, n_global=8, comm_stream=comm)
        assert np.isfinite(float(loss2)) and not step_aborted(m)
        assert not step_skipped_nonfinite(m, opt, float(loss2))
        for p, w in zip(m.parameters(), want):
            assert torch.equal(p.detach(), w)
        assert all(int(st["step"]) == 1 for st in opt.state.values())
    finally:
        lib.opnet_xcd4_enable(1)
        lib.opseq_xcd_enable(1)          # (a data-parallel abort takes every rank off BOTH persistent kernel families)


def test_non_finite_loss_is_skipped_on_every_rank_and_reported(rccl_world_of_one, monkeypatch):
    """a NaN loss no abort word announces - the decision comes from the all-reduced guard slot 1 (every rank
    alike), step_skipped_nonfinite() reports it and rolls the step counters back"""
    from objectpermanence_amd import FusedAdam, parallel
    from objectpermanence_amd.training import step_aborted, step_skipped_nonfinite, train_step
    monkeypatch.setenv(parallel.FORCE_ENV, "1")
    m = _model()
    opt = FusedAdam(m.parameters(), lr=1e-3)
    boxes_np, labels_np = synth.make_batch(3, 4, 6)
    good = torch.from_numpy(labels_np).to("cuda:0")
    bad_np = labels_np.copy()
    bad_np[0, 0, 0] = np.nan
    boxes = torch.from_numpy(boxes_np).to("cuda:0")
    train_step("opnet", m, opt, boxes, good, n_global=4)
    before = [p.detach().clone() for p in m.parameters()]
    loss = train_step("opnet", m, opt, boxes, torch.from_numpy(bad_np).to("cuda:0"), n_global=4)
    assert not np.isfinite(float(loss))
    g = m._grad_bucket.guard.cpu().numpy()
    assert g[0] == 0.0 and g[1] == 1.0
    for p, b in zip(m.parameters(), before):
        assert torch.equal(p.detach(), b)
    assert not step_aborted(m)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert step_skipped_nonfinite(m, opt, 0.0)           # (the local loss of ANOTHER rank may well be finite)
    assert all(int(st["step"]) == 1 for st in opt.state.values())
    train_step("opnet", m, opt, boxes, good, n_global=4)
    assert all(int(st["step"]) == 2 for st in opt.state.values()) and not step_skipped_nonfinite(m, opt, 1.0)


def _write_videos(tmp_path, tag, n, first, with_mask=False):
    s, l = tmp_path / f"{tag}_s", tmp_path / f"{tag}_l"
    s.mkdir(); l.mkdir()
    lines = []
    for i in range(n):
        name = f"{tag}{i:02d}"
        bb, lab, gt = synth.make_raw_video(first + i, "plain")
        pickle.dump({"bb": bb, "labels": lab}, open(s / (name + ".pkl"), "wb"), pickle.HIGHEST_PROTOCOL)
        json.dump(gt, open(l / (name + "_bb.json"), "w"))
        lines.append(name + "\t" + ",".join(str(x) for x in range(10 * i, 10 * i + 25)) + "\n")
    if with_mask:
        open(tmp_path / f"{tag}_mask.txt", "w").writelines(lines)
    return str(s), str(l), str(tmp_path / f"{tag}_mask.txt")


def _inference_files(tmp_path, n=7, device="cuda:0", batch_size=3):
    s, l, _ = _write_videos(tmp_path, "v", n, 40)
    params = synth.opnet_synth_params(CFG)
    torch.save({k: torch.from_numpy(v) for k, v in params.items()}, tmp_path / "opnet.pth")
    json.dump(CFG, open(tmp_path / "model.json", "w"))
    json.dump({"batch_size": batch_size, "num_workers": 0, "device": device, "model_path": str(tmp_path / "opnet.pth"),
               "videos_dir": "unused", "sample_dir": s, "labels_dir": l}, open(tmp_path / "infer.json", "w"))
    return str(tmp_path / "infer.json"), str(tmp_path / "model.json")


def test_inference_driver_under_the_group_equals_the_plain_run(rccl_world_of_one, monkeypatch, tmp_path):
    from objectpermanence_amd import parallel
    from objectpermanence_amd.inference_main import reasoning_inference_main
    infer, model = _inference_files(tmp_path)
    monkeypatch.delenv(parallel.FORCE_ENV, raising=False)
    plain = reasoning_inference_main("opnet", str(tmp_path / "out0"), infer, model)
    monkeypatch.setenv(parallel.FORCE_ENV, "1")
    calls = []
    real = parallel.all_gather_by_index
    monkeypatch.setattr(parallel, "all_gather_by_index", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    dp = reasoning_inference_main("opnet", str(tmp_path / "out1"), infer, model)
    assert len(calls) == 2                                 # predictions and IoUs went through the RCCL gathers
    assert dp["video_names"] == plain["video_names"] and np.array_equal(dp["predictions"], plain["predictions"])
    assert dp["mean_iou"] == plain["mean_iou"] and dp["map_0.5"] == plain["map_0.5"]
    for n in plain["video_names"]:
        assert open(tmp_path / "out0" / (n + "_bb.json")).read() == open(tmp_path / "out1" / (n + "_bb.json")).read()


def test_training_driver_under_the_group_equals_the_plain_run(rccl_world_of_one, monkeypatch, tmp_path):
    """training_main (train steps + the sharded per-epoch evaluation + checkpoint) under the group: same history, same
    checkpointed weights"""
    from objectpermanence_amd import parallel
    from objectpermanence_amd.training_main import training_main
    tr = _write_videos(tmp_path, "train", 10, 100, with_mask=True)
    dv = _write_videos(tmp_path, "dev", 3, 100, with_mask=True)

    def run(tag):
        cfg = {"batch_size": 4, "inference_batch_size": 400, "num_workers": 0, "num_epochs": 2, "print_step": 100,
               "learning_rate": 0.001, "lr_scheduler_patience": 2, "lr_scheduler_factor": 0.8, "device": "cuda:0",
               "checkpoints_path": str(tmp_path / f"ckpt_{tag}"),
               "train_sample_dir": tr[0], "train_labels_dir": tr[1], "train_containment_file": tr[2],
               "dev_sample_dir": dv[0], "dev_labels_dir": dv[1], "dev_containment_file": dv[2]}
        torch.manual_seed(0)
        return training_main("opnet", cfg, CFG)

    monkeypatch.delenv(parallel.FORCE_ENV, raising=False)
    plain = run("plain")
    monkeypatch.setenv(parallel.FORCE_ENV, "1")
    dp = run("dp")
    assert plain["history"] == dp["history"]
    if plain["checkpoint"]:
        a, b = torch.load(plain["checkpoint"]), torch.load(dp["checkpoint"])
        assert set(a) == set(b) and all(torch.equal(a[k], b[k]) for k in a)


def _torchrun(args, tmp_path, extra_env=None, nproc=1):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = REPO + os.pathsep + env.get("PYTHONPATH", "")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(29900 + os.getpid() % 90), "-m", "objectpermanence_amd"] + args
    return subprocess.run(cmd, cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)


def test_torchrun_entry_point_overrides_the_json_device(tmp_path):
    """the documented launch line.  The JSON names cuda:7 - a device this box does not have (and the device all eight ranks of a
    real job would otherwise share): the rank runs on cuda:LOCAL_RANK, joins RCCL, gathers, and rank 0 writes the files"""
    from objectpermanence_amd.datasets import encode_boxes
    infer, model = _inference_files(tmp_path, n=5, device="cuda:7")
    r = _torchrun(["inference", "--model_type", "opnet", "--results_dir", str(tmp_path / "out"), "--inference_config", infer,
                   "--model_config", model], tmp_path, {"OPNET_FORCE_DIST": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    params = synth.opnet_synth_params(CFG)
    names = sorted(p.stem for p in (tmp_path / "v_s").glob("*.pkl"))
    raws = [pickle.load(open(tmp_path / "v_s" / (n + ".pkl"), "rb")) for n in names]
    boxes = np.stack([encode_boxes(r_["bb"], r_["labels"], 6).astype(np.float32) for r_ in raws])
    px = oo.postprocess_to_pixels(oo.opnet_forward(boxes, params, np.float32)[0])
    for n, p in zip(names, px):
        got = np.array(json.load(open(tmp_path / "out" / (n + "_bb.json"))))
        assert got.shape == p.shape and np.abs(got - p).max() <= 1 and (got != p).mean() < 2e-3


def test_bench_train_force_dist_reports_the_allreduce(tmp_path):
    """bench.py --mode train --force-dist = config 5's line on one GPU: the collective's own time must be in it"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MASTER_PORT"] = str(29800 + os.getpid() % 90)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--mode", "train", "--force-dist", "--steps", "3",
                        "--warmup", "1", "--repeats", "2", "--no-cpu-baseline"], cwd=REPO, env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["allreduce_ms_per_step"] is not None and 0 < line["allreduce_ms_per_step"] < 5.0
    assert line["engine"] == "xcd4" and np.isfinite(line["final_loss"])
    c = line["collective"]                               # the line proves its own exchange: backend, ranks one all-reduce saw, its time
    assert c["backend"] == "nccl" and c["ranks_seen"] == c["world_size"] == 1 and c["allreduce_ms_per_step"] == line["allreduce_ms_per_step"]


@pytest.mark.parametrize("overlap", ["0", "1"])
def test_bench_inference_force_dist_reports_the_allgather(tmp_path, overlap):
    """the headline command with --force-dist = the N > 1 inference line on one GPU (RCCL group of one): the persistent engine, one
    all-gather per launch on the side stream, and the `collective` block that answers "did RCCL see N ranks" from the line itself;
    OPNET_DP_OVERLAP=1 lets the launch start without waiting for the previous gather (the A/B switch for the first multi-GPU run)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["MASTER_PORT"] = str(29700 + os.getpid() % 90)
    env["OPNET_DP_OVERLAP"] = overlap
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--force-dist", "--steps", "8", "--warmup", "2", "--repeats", "3",
                        "--inflight", "4", "--no-cpu-baseline"], cwd=REPO, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    c = line["collective"]
    assert c["backend"] == "nccl" and c["ranks_seen"] == c["world_size"] == 1
    assert c["allgathers"] == 2 * 3 and 0 < c["allgather_ms_per_launch"] < 5.0 and c["bytes_per_rank_per_launch"] == 4 * 32 * 300 * 4 * 4
    assert c["launch_waits_for_gather"] == (overlap == "0")
    assert line["config"]["engine"] == "xcd" and line["value"] > 0
