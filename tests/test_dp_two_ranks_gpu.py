"""BASELINE config 5 at world size TWO with the HIP model, on the one GPU a test box has: two processes (LOCAL_RANK 0 both, gloo for
the collectives - RCCL refuses two ranks on one device) each run the product's `train_step` on their contiguous half of every
minibatch: in-place gradient bucket, all-reduce weighted n_local / n_global, guard slots, guarded Adam.  The two ranks must end
with IDENTICAL weights, and those must be the single-process full-minibatch weights up to the order of the fp32 sums (the reference's
contract for a sharded run: N-GPU == 1-GPU, training_main.py:183-217).  The persistent engines are switched off in the workers (two
processes cannot both hold all 256 CUs): what runs is the launch chain, the same arithmetic behind the same exchange.
tests/test_dp_rccl_gpu.py has the RCCL transport (world of one, bit-identical); test_host_logic.py the gloo logic on CPU."""
import json
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import synth

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> str:
    """a port nobody listens on right now (every launch gets its own: a rendezvous port in TIME_WAIT must not fail the next test)"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return str(sk.getsockname()[1])
CFG = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}


def _single(B, T, steps):
    from objectpermanence_amd import FusedAdam, ModelsFactory
    from objectpermanence_amd.training import train_step
    m = ModelsFactory.get_model("opnet", CFG)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.opnet_synth_params(CFG).items()})
    m = m.to("cuda:0").train(True)
    opt = FusedAdam(m.parameters(), lr=1e-3)
    losses, grads1 = [], None
    for k in range(steps):
        boxes_np, labels_np = synth.make_batch(10 * k, B, T)
        losses.append(float(train_step("opnet", m, opt, torch.from_numpy(boxes_np).to("cuda:0"), torch.from_numpy(labels_np).to("cuda:0"),
                                       n_global=B)))
        if k == 0:
            grads1 = {n: p.grad.detach().cpu().clone() for n, p in m.named_parameters()}
    torch.cuda.synchronize()
    return {n: p.detach().cpu() for n, p in m.named_parameters()}, losses, grads1


@pytest.mark.parametrize("B,T", [(12, 10), (7, 25)])          # (7, 25): shards of 4 and 3 clips - unequal weights n_local / n_global
def test_two_ranks_on_one_gpu_train_to_the_single_process_weights(tmp_path, B, T):
    steps = 3
    env = dict(os.environ, WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(),
               OPNET_XCD4="0", OPNET_XCD="0", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=REPO)
    env.pop("OPNET_FORCE_DIST", None)
    procs = [subprocess.Popen([sys.executable, os.path.join(REPO, "tests", "_dp2_worker.py"), str(tmp_path), str(B), str(T), str(steps)],
                              env=dict(env, RANK=str(r)), cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=600)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(2))
    want, want_losses, want_grads = _single(B, T, steps)
    gmax = max(float(g.abs().max()) for g in want_grads.values())
    for n in want:
        assert torch.equal(r0["params"][n], r1["params"][n]), n                     # the ranks agree bit for bit
        assert torch.equal(r0["grads1"][n], r1["grads1"][n]), n
        # the exchanged gradient is the single-process mean-loss gradient up to the order of the fp32 sums ...
        assert float((r0["grads1"][n] - want_grads[n]).abs().max()) <= 2e-5 * gmax, n
        # ... and three Adam steps (lr 1e-3: an update of ~1e-3 per step whatever the gradient's size, so rounding noise in
        # near-zero gradients shows at the 1e-5 level) land within a percent of one step's size
        assert float((r0["params"][n] - want[n]).abs().max()) <= 3e-5, n
    for k in range(steps):
        # slot 2 of the guard carries the whole minibatch's loss to every rank: sum of the shards' n_local / n_global shares
        assert r0["losses"][k][1] == r1["losses"][k][1]
        assert abs(r0["losses"][k][1] - want_losses[k]) <= 1e-6 * max(1.0, abs(want_losses[k]))
        if B % 2 == 0:
            assert r0["losses"][k][0] != r1["losses"][k][0]                          # (their own shards' losses differ)
    assert float(r0["guard"][0]) == 0.0 and float(r0["guard"][1]) == 0.0


def _two_ranks(args, tmp_path, seeds=("0", "0")):
    """`python -m objectpermanence_amd <args>` as two ranks on cuda:0 (what torchrun would start, but LOCAL_RANK 0 twice)"""
    env = dict(os.environ, WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(),
               OPNET_DIST_BACKEND="gloo", OPNET_XCD4="0", OPNET_XCD="0", OPSEQ_XCD="0", PYTHONPATH=REPO)
    env.pop("OPNET_FORCE_DIST", None)
    procs = [subprocess.Popen([sys.executable, "-m", "objectpermanence_amd"] + args, env=dict(env, RANK=str(r), OPNET_SEED=seeds[r]), cwd=str(tmp_path),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=900)[0].decode(errors="replace") for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(o[-3000:] for o in outs)
    return outs


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


def test_inference_entry_point_at_world_size_two_writes_the_single_process_files(tmp_path, monkeypatch):
    """`python -m objectpermanence_amd inference` as two ranks: clips sharded 4 + 3, predictions and IoUs gathered by dataset index,
    rank 0 alone writes - the files equal the single-process run's byte for byte (same engine: clips are independent)"""
    from objectpermanence_amd.inference_main import reasoning_inference_main
    s, l, _ = _write_videos(tmp_path, "v", 7, 40)
    torch.save({k: torch.from_numpy(v) for k, v in synth.opnet_synth_params(CFG).items()}, tmp_path / "opnet.pth")
    json.dump(CFG, open(tmp_path / "model.json", "w"))
    json.dump({"batch_size": 3, "num_workers": 0, "device": "cuda:0", "model_path": str(tmp_path / "opnet.pth"), "videos_dir": "unused",
               "sample_dir": s, "labels_dir": l}, open(tmp_path / "infer.json", "w"))
    _two_ranks(["inference", "--model_type", "opnet", "--results_dir", str(tmp_path / "out2"), "--inference_config",
                str(tmp_path / "infer.json"), "--model_config", str(tmp_path / "model.json")], tmp_path)
    monkeypatch.setenv("OPNET_XCD4", "0")
    monkeypatch.setenv("OPNET_XCD", "0")
    plain = reasoning_inference_main("opnet", str(tmp_path / "out1"), str(tmp_path / "infer.json"), str(tmp_path / "model.json"))
    assert len(plain["video_names"]) == 7
    for n in plain["video_names"]:
        assert open(tmp_path / "out1" / (n + "_bb.json")).read() == open(tmp_path / "out2" / (n + "_bb.json")).read(), n


def test_training_entry_point_at_world_size_two_checkpoints_the_single_process_weights(tmp_path, monkeypatch):
    """`python -m objectpermanence_amd training` as two ranks: every minibatch of 4 split 2 + 2, gradients all-reduced, the
    per-epoch evaluation sharded and gathered, rank 0 alone checkpoints - the weights are the single-process run's up to the
    order of the fp32 sums through six Adam steps.  The ranks are seeded DIFFERENTLY (0 and 123): rank 1 must take rank 0's random
    initialisation (parallel.broadcast_parameters), as N processes of the reference's unseeded construction would differ"""
    from objectpermanence_amd.training_main import training_main
    tr = _write_videos(tmp_path, "train", 8, 100, with_mask=True)
    dv = _write_videos(tmp_path, "dev", 3, 100, with_mask=True)

    def cfg(tag):
        return {"batch_size": 4, "inference_batch_size": 400, "num_workers": 0, "num_epochs": 3, "print_step": 100,
                "learning_rate": 0.001, "lr_scheduler_patience": 2, "lr_scheduler_factor": 0.8, "device": "cuda:0",
                "checkpoints_path": str(tmp_path / f"ckpt_{tag}"),
                "train_sample_dir": tr[0], "train_labels_dir": tr[1], "train_containment_file": tr[2],
                "dev_sample_dir": dv[0], "dev_labels_dir": dv[1], "dev_containment_file": dv[2]}

    json.dump(CFG, open(tmp_path / "model.json", "w"))
    json.dump(cfg("dp"), open(tmp_path / "train.json", "w"))
    _two_ranks(["training", "--model_type", "opnet", "--model_config", str(tmp_path / "model.json"), "--training_config",
                str(tmp_path / "train.json")], tmp_path, seeds=("0", "123"))
    monkeypatch.setenv("OPNET_XCD4", "0")
    monkeypatch.setenv("OPNET_XCD", "0")
    torch.manual_seed(0)
    plain = training_main("opnet", cfg("plain"), CFG)
    got = sorted((tmp_path / "ckpt_dp").rglob("*.pth"))
    want = sorted((tmp_path / "ckpt_plain").rglob("*.pth"))
    assert want and [p.name for p in got] == [p.name for p in want]      # same epochs checkpointed (same dev-loss history), by rank 0 only
    a, b = torch.load(got[-1]), torch.load(want[-1])
    assert set(a) == set(b)
    for k in a:
        assert float((a[k].float() - b[k].float()).abs().max()) <= 1e-4, k


@pytest.mark.parametrize("mode_args", [["--engine", "chain"], ["--mode", "train"], ["--mode", "train", "--global-batch", "8"]])
def test_bench_as_two_ranks_prints_one_line_from_rank_zero(tmp_path, mode_args):
    """the driver's `--gpus N` command, rehearsed at N = 2 on the one GPU (gloo, launch-chain engines): the barrier / max-over-ranks
    timing, the periodic all-gather of predictions (inference) or the bucket all-reduce (training) run, rank 0 prints exactly one JSON
    line with n_gpus 2 and dp2, rank 1 prints nothing"""
    env = dict(os.environ, WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=_free_port(),
               OPNET_DIST_BACKEND="gloo", OPNET_XCD4="0", OPNET_XCD="0", PYTHONPATH=REPO)
    env.pop("OPNET_FORCE_DIST", None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--repeats", "2",
           "--no-cpu-baseline"] + mode_args
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r)), cwd=REPO, stdout=subprocess.PIPE, stderr=subprocess.PIPE) for r in range(2)]
    res = [p.communicate(timeout=900) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(e.decode(errors="replace")[-2000:] for _, e in res)
    out0, out1 = res[0][0].decode().strip(), res[1][0].decode().strip()
    assert out1 == "" and len(out0.splitlines()) == 1
    line = json.loads(out0)
    assert line["n_gpus"] == 2 and line["config"]["parallelism"] == "dp2"
    if "--global-batch" in mode_args:       # BASELINE.json config 5 at its FIXED global batch: per-rank batch = global / N
        assert line["scaling"] == "strong" and line["config"]["global_batch"] == 8 and "batch=4 clips/GPU" in line["config"]["workload"]
    else:
        assert line["scaling"] == "weak"
    assert line["value"] > 0 and line["steps"] == 3 and line["warmup"] == 1
    # the line answers "did the collective see N ranks" itself
    assert line["collective"]["backend"] == "gloo" and line["collective"]["ranks_seen"] == line["collective"]["world_size"] == 2
    if "train" in mode_args:
        assert line["allreduce_ms_per_step"] is not None and np.isfinite(line["final_loss"])
        assert line["collective"]["allreduce_ms_per_step"] == line["allreduce_ms_per_step"]
    else:
        assert line["collective"]["launch_waits_for_gather"] is False


def test_cater_inference_entry_point_at_world_size_two_writes_the_single_process_csv(tmp_path, monkeypatch):
    """`python -m objectpermanence_amd cater_inference` as two ranks: minibatches dealt to the ranks, last-frame boxes gathered by
    dataset index, rank 0 writes class_pred_results.csv - the same file as the single-process run"""
    from objectpermanence_amd.cater_setup_inference import cater_setup_inference
    s, l, _ = _write_videos(tmp_path, "CATER_new_0000", 7, 40)
    torch.save({k: torch.from_numpy(v) for k, v in synth.opnet_synth_params(CFG).items()}, tmp_path / "opnet.pth")
    json.dump(CFG, open(tmp_path / "model.json", "w"))
    json.dump({"batch_size": 2, "num_workers": 0, "device": "cuda:0", "model_path": str(tmp_path / "opnet.pth"), "sample_dir": s,
               "labels_dir": l}, open(tmp_path / "infer.json", "w"))
    _two_ranks(["cater_inference", "--results_dir", str(tmp_path / "out2"), "--inference_config", str(tmp_path / "infer.json"),
                "--model_config", str(tmp_path / "model.json")], tmp_path)
    monkeypatch.setenv("OPNET_XCD4", "0")
    monkeypatch.setenv("OPNET_XCD", "0")
    df = cater_setup_inference("opnet", str(tmp_path / "out1"), str(tmp_path / "infer.json"), str(tmp_path / "model.json"))
    assert len(df) == 7
    assert open(tmp_path / "out1" / "class_pred_results.csv").read() == open(tmp_path / "out2" / "class_pred_results.csv").read()


def test_transformer_inference_at_world_size_two_keeps_the_reference_minibatches(tmp_path, monkeypatch):
    """transformer_lstm couples the clips of a minibatch (attention over B x T tokens): the ranks take WHOLE reference minibatches
    (7 clips at batch_size 2: minibatches 0, 2 / 1, 3), never a cut through one - the files equal the single-process run's"""
    from objectpermanence_amd.inference_main import reasoning_inference_main
    cfg = {"boxes_features_dim": 256, "num_attention_heads": 4, "num_attention_layers": 2, "num_lstm_layers": 2, "lstm_hidden_dim": 512}
    s, l, _ = _write_videos(tmp_path, "v", 7, 60)
    torch.save({k: torch.from_numpy(v) for k, v in synth.transformer_lstm_synth_params(cfg).items()}, tmp_path / "t.pth")
    json.dump(cfg, open(tmp_path / "model.json", "w"))
    json.dump({"batch_size": 2, "num_workers": 0, "device": "cuda:0", "model_path": str(tmp_path / "t.pth"), "videos_dir": "unused",
               "sample_dir": s, "labels_dir": l}, open(tmp_path / "infer.json", "w"))
    _two_ranks(["inference", "--model_type", "transformer_lstm", "--results_dir", str(tmp_path / "out2"), "--inference_config",
                str(tmp_path / "infer.json"), "--model_config", str(tmp_path / "model.json")], tmp_path)
    monkeypatch.setenv("OPSEQ_XCD", "0")
    plain = reasoning_inference_main("transformer_lstm", str(tmp_path / "out1"), str(tmp_path / "infer.json"), str(tmp_path / "model.json"))
    assert len(plain["video_names"]) == 7
    for n in plain["video_names"]:
        assert open(tmp_path / "out1" / (n + "_bb.json")).read() == open(tmp_path / "out2" / (n + "_bb.json")).read(), n


def test_preprocess_entry_point_at_world_size_two_deals_the_videos(tmp_path):
    """`python -m objectpermanence_amd preprocess` as two ranks: videos dealt round-robin (no collective), every rank writes the
    <video>.pkl files of its own videos into the shared directory - together exactly the single-process run's files"""
    from oracle import detector_oracle as do
    from objectpermanence_amd.preprocess_perception_main import preprocess_main
    vids = tmp_path / "videos"
    vids.mkdir()
    rng = np.random.default_rng(3)
    for i in range(3):
        np.save(vids / f"cater_{i:03d}.npy", rng.integers(0, 256, size=(300, 60, 80, 3), dtype=np.uint8))
    np.save(vids / "cater_short.npy", rng.integers(0, 256, size=(299, 60, 80, 3), dtype=np.uint8))      # not 300 frames: never written
    sd = {k: torch.from_numpy(v) for k, v in {**do.synth_backbone_params(), **do.synth_head_params()}.items()}
    torch.save({"model_state_dict": sd}, tmp_path / "detection_model.pth")
    json.dump({"videos_dir": str(vids), "od_model_weights": str(tmp_path / "detection_model.pth"), "device": "cuda:0"},
              open(tmp_path / "preprocess.json", "w"))
    (tmp_path / "res2").mkdir()
    (tmp_path / "res1").mkdir()
    _two_ranks(["preprocess", "--results_dir", str(tmp_path / "res2"), "--config", str(tmp_path / "preprocess.json")], tmp_path)
    assert preprocess_main(str(tmp_path / "res1"), str(tmp_path / "preprocess.json")) == 3
    names = sorted(os.listdir(tmp_path / "res1"))
    assert names == sorted(os.listdir(tmp_path / "res2")) == [f"cater_{i:03d}.pkl" for i in range(3)]
    for n in names:
        a, b = pickle.load(open(tmp_path / "res1" / n, "rb")), pickle.load(open(tmp_path / "res2" / n, "rb"))
        assert all(np.array_equal(x, y) for x, y in zip(a["bb"], b["bb"])) and all(np.array_equal(x, y) for x, y in zip(a["labels"], b["labels"]))


def test_transformer_training_entry_point_runs_at_world_size_two(tmp_path):
    """`training --model_type transformer_lstm` as two ranks (whole reference minibatches per rank, dropout live, the encoder's
    gradients through the same bucket): it must finish - an uneven number of minibatches per rank (3 minibatches, 2 ranks) once hung
    the all-reduce on CPU (test_host_logic) - and rank 0 alone checkpoints finite weights"""
    tr = _write_videos(tmp_path, "train", 6, 100, with_mask=True)
    dv = _write_videos(tmp_path, "dev", 2, 100, with_mask=True)
    cfg = {"boxes_features_dim": 256, "num_attention_heads": 4, "num_attention_layers": 2, "num_lstm_layers": 2, "lstm_hidden_dim": 512}
    json.dump(cfg, open(tmp_path / "model.json", "w"))
    json.dump({"batch_size": 2, "inference_batch_size": 2, "num_workers": 0, "num_epochs": 2, "print_step": 100, "learning_rate": 0.001,
               "lr_scheduler_patience": 2, "lr_scheduler_factor": 0.8, "device": "cuda:0", "checkpoints_path": str(tmp_path / "ckpt"),
               "train_sample_dir": tr[0], "train_labels_dir": tr[1], "train_containment_file": tr[2],
               "dev_sample_dir": dv[0], "dev_labels_dir": dv[1], "dev_containment_file": dv[2]}, open(tmp_path / "train.json", "w"))
    outs = _two_ranks(["training", "--model_type", "transformer_lstm", "--model_config", str(tmp_path / "model.json"), "--training_config",
                       str(tmp_path / "train.json")], tmp_path, seeds=("0", "5"))
    assert "Epoch 2 Dev Set" in outs[0] and "Epoch" not in outs[1]                  # rank 0 alone reports
    ck = sorted((tmp_path / "ckpt").rglob("*.pth"))
    assert ck
    sd = torch.load(ck[-1])
    assert all(bool(torch.isfinite(v.float()).all()) for v in sd.values())
