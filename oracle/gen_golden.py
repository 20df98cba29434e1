#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING THE REFERENCE in the build container.

Runs only where /root/reference exists (never on the GPU box, never from tests).
It imports the reference's own model classes (baselines/learned_models.py) and
metric class (baselines/tracking_utils.py:ResultsAnalyzer) under the container's
torch 2.10 CPU, feeds them the deterministic synthetic weights / clips of
oracle/synth.py, and stores inputs-free, outputs-only fixtures (inputs are
regenerated from seeds by the tests).

    python oracle/gen_golden.py            # rewrites tests/golden/
"""
from __future__ import annotations

import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("OPNET_REFERENCE", "/root/reference")
# OPNET_GOLDEN_OUT: write somewhere else (tests/test_golden_regeneration.py regenerates into a temp dir and compares)
OUT = os.environ.get("OPNET_GOLDEN_OUT") or os.path.join(REPO, "tests", "golden")

sys.path.insert(0, REPO)
from oracle import synth  # noqa: E402


def _import_reference():
    # the reference imports cv2 / torchvision at module top (tracking_utils.py:7, detector.py:6-8);
    # neither is installed here and neither is on the reasoner path -> stub modules.
    for name in ("cv2", "torchvision", "torchvision.models", "torchvision.models.detection",
                 "torchvision.models.detection.faster_rcnn"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision.models.detection.faster_rcnn"].FastRCNNPredictor = object
    if not hasattr(np, "int"):
        np.int = int  # removed numpy aliases used at tracking_utils.py:270,272
    if not hasattr(np, "bool"):
        np.bool = bool
    sys.path.insert(0, REF)
    from baselines import learned_models  # noqa
    from baselines import tracking_utils  # noqa
    return learned_models, tracking_utils


def _load_params(model: torch.nn.Module, params):
    sd = model.state_dict()
    assert set(sd.keys()) == set(params.keys()), (sorted(sd.keys()), sorted(params.keys()))
    for k, v in params.items():
        assert tuple(sd[k].shape) == tuple(v.shape), (k, sd[k].shape, v.shape)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})


def gen_opnet(lm, cfg, n_clips, t_frames, tag, keep_intermediates):
    params = synth.opnet_synth_params(cfg)
    model = lm.OPNet(cfg)
    _load_params(model, params)
    model.eval()
    boxes, labels = synth.make_batch(0, n_clips, t_frames)
    with torch.no_grad():
        y, logits = model(torch.from_numpy(boxes))
    out = {"y": y.numpy(), "logits": logits.numpy(),
           "cfg": np.array(json.dumps(cfg)), "n_clips": n_clips, "t_frames": t_frames}
    if keep_intermediates:
        with torch.no_grad():
            scene = torch.from_numpy(boxes).view(n_clips, t_frames, -1)
            h1, _ = model.object_to_track_LSTM(scene)
            probs = torch.softmax(model.object_to_track_prediction(h1), dim=-1)
            fb = torch.einsum("bfot,bfo->bft", torch.from_numpy(boxes), probs)
            h2, _ = model.video_LSTM(fb)
        out.update(h1=h1.numpy(), probs=probs.numpy(), frames_boxes=fb.numpy(), h2=h2.numpy())
    else:
        # a few checksums of intermediates at the real size
        with torch.no_grad():
            scene = torch.from_numpy(boxes).view(n_clips, t_frames, -1)
            h1, _ = model.object_to_track_LSTM(scene)
        out["h1_last"] = h1[:, -1].numpy()
    # batch independence: clip 0 alone == clip 0 in the batch
    with torch.no_grad():
        y0, _ = model(torch.from_numpy(boxes[:1]))
    out["y_clip0_alone"] = y0.numpy()
    np.savez_compressed(os.path.join(OUT, f"opnet_{tag}.npz"), **out)
    print(f"opnet_{tag}: y range [{y.min():.3f}, {y.max():.3f}] logits range "
          f"[{logits.min():.2f}, {logits.max():.2f}]")
    return y.numpy(), labels


def gen_siblings(lm):
    """forward goldens of BaselineLstm, NonLinearLstm, OPNetLstmMlp, TransformerLstm (learned_models.py:55-197)"""
    torch.manual_seed(0)
    out = {}
    cases = {
        "baseline_lstm": (lm.BaselineLstm, synth.baseline_lstm_synth_params,
                          [("tiny", {"videos_hidden_dim": 32}, 2, 10), ("real", {"videos_hidden_dim": 512}, 3, 300)]),
        "non_linear_lstm": (lm.NonLinearLstm, synth.non_linear_lstm_synth_params,
                            [("tiny", {"boxes_features_dim": 16, "videos_hidden_dim": 32}, 2, 10),
                             ("real", {"boxes_features_dim": 256, "videos_hidden_dim": 512}, 2, 60)]),
        "opnet_lstm_mlp": (lm.OPNetLstmMlp, synth.opnet_lstm_mlp_synth_params,
                           [("tiny", {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 16, "videos_hidden_dim": 32}, 2, 10),
                            ("real", {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}, 3, 300)]),
    }
    for name, (cls, pfn, variants) in cases.items():
        for tag, cfg, n, t in variants:
            model = cls(cfg)
            _load_params(model, pfn(cfg))
            model.eval()
            boxes, _ = synth.make_batch(0, n, t)
            x = boxes if name == "opnet_lstm_mlp" else synth.boxes5(boxes)
            with torch.no_grad():
                y = model(torch.from_numpy(x))
            if isinstance(y, tuple):
                out[f"{name}/{tag}/logits"] = y[1].numpy()
                y = y[0]
            out[f"{name}/{tag}/y"] = y.numpy()
            out[f"{name}/{tag}/cfg"] = np.array(json.dumps(cfg))
            out[f"{name}/{tag}/shape"] = np.array([n, t])
            print(f"{name}/{tag}: y range [{float(y.min()):.3f}, {float(y.max()):.3f}]")
    # transformer_lstm: tiny, the JSON config (2 heads) at B=1 and B=2 (batch coupling), BASELINE's 4 heads
    tcases = [("tiny", {"boxes_features_dim": 32, "num_attention_heads": 2, "num_attention_layers": 2,
                        "num_lstm_layers": 2, "lstm_hidden_dim": 32}, 2, 6),
              ("real_b1", None, 1, 300), ("real_b2", None, 2, 300), ("heads4_b1", "h4", 1, 300)]
    with open(os.path.join(REF, "configs", "transformer_lstm_model_config.json")) as f:
        real = json.load(f)
    for tag, cfg, n, t in tcases:
        if cfg is None:
            cfg = dict(real)
        elif cfg == "h4":
            cfg = dict(real); cfg["num_attention_heads"] = 4
        model = lm.TransformerLstm(cfg)
        _load_params(model, synth.transformer_lstm_synth_params(cfg))
        model.eval()
        boxes, _ = synth.make_batch(0, n, t)
        with torch.no_grad():
            y = model(torch.from_numpy(synth.boxes5(boxes)))
        out[f"transformer_lstm/{tag}/y"] = y.numpy()
        out[f"transformer_lstm/{tag}/cfg"] = np.array(json.dumps(cfg))
        out[f"transformer_lstm/{tag}/shape"] = np.array([n, t])
        print(f"transformer_lstm/{tag}: y range [{float(y.min()):.3f}, {float(y.max()):.3f}]")
    np.savez_compressed(os.path.join(OUT, "siblings.npz"), **out)


def gen_datasets():
    """Dataset-encode goldens: write synthetic <video>.pkl / <video>_bb.json files and run the
    reference's own Cater{5,6}TracksForObjectsInferenceDataset.__getitem__ and the training variant's mask."""
    import pickle
    import tempfile
    sys.path.insert(0, REF)
    from baselines import datasets as rd
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        samples, labels_dir = os.path.join(tmp, "s"), os.path.join(tmp, "l")
        os.makedirs(samples); os.makedirs(labels_dir)
        variants = ["plain", "dups", "crowded", "nosnitch0", "sparse"]
        lines = []
        for i, v in enumerate(variants):
            name = f"vid_{i:02d}_{v}"
            bb, lab, gt = synth.make_raw_video(i, v)
            with open(os.path.join(samples, name + ".pkl"), "wb") as f:
                pickle.dump({"bb": bb, "labels": lab}, f, pickle.HIGHEST_PROTOCOL)
            with open(os.path.join(labels_dir, name + "_bb.json"), "w") as f:
                json.dump(gt, f)
            frames = sorted(set(int(x) for x in np.random.default_rng(i).integers(0, 300, size=20 * i)))
            lines.append(name + "\t" + ",".join(str(x) for x in frames) + "\n")
        mask_file = os.path.join(tmp, "containment.txt")
        with open(mask_file, "w") as f:
            f.writelines(lines)
        for tracks, cls_inf, cls_tr in ((6, rd.Cater6TracksForObjectsInferenceDataset, rd.Cater6TracksForObjectsTrainingDataset),
                                        (5, rd.Cater5TracksForObjectsInferenceDataset, rd.Cater5TracksForObjectsTrainingDataset)):
            ds = cls_inf(samples, labels_dir)
            dt = cls_tr(samples, labels_dir, mask_file)
            assert len(ds) == len(variants)
            for i in range(len(ds)):
                (boxes, idx), (lab, _), name = ds[i]
                out[f"t{tracks}/{i}/name"] = np.array(name)
                out[f"t{tracks}/{i}/boxes"] = boxes.numpy()
                out[f"t{tracks}/{i}/index"] = idx.numpy()
                out[f"t{tracks}/{i}/labels"] = lab.numpy()
                (_, _), (_, mask), _ = dt[i]
                out[f"t{tracks}/{i}/mask"] = mask.numpy()
    np.savez_compressed(os.path.join(OUT, "datasets.npz"), **out)
    print("datasets: ", {k: out[k].shape for k in list(out)[:5]})


def gen_trained(lm, tu, cfg):
    """Reference forward + ResultsAnalyzer metrics with TRAINED weights (tests/golden/opnet_trained_fp16.npz:
    OPNet trained for 40 epochs on synthetic clips by tools/train_synthetic.py on the MI355X with this repo's
    own training path, stored rounded to fp16) on held-out synthetic clips: a non-vacuous mean-IoU / mAP@0.5
    parity target (north_star: mean-IoU within 1e-3)."""
    path = os.path.join(OUT, "opnet_trained_fp16.npz")
    if not os.path.exists(path):
        print("gen_trained: no trained weights fixture, skipped")
        return
    w = np.load(path)
    params = {k: w[k].astype(np.float32) for k in w.files}
    model = lm.OPNet(cfg)
    _load_params(model, params)
    model.eval()
    first, n = 200000, 16
    boxes, labels = synth.make_batch(first, n, 300)
    with torch.no_grad():
        y, _ = model(torch.from_numpy(boxes))
    y = y.numpy()
    fs = np.array([320, 240, 320, 240])
    pred_px = (np.array(list(y.reshape(-1, 4))) * fs).reshape((n, 300, 4)).astype(np.int32)       # inference_main.py:219
    gt_px = (np.array(list(labels.reshape(-1, 4))) * fs).reshape((n, 300, 4)).astype(np.int32)
    names = [str(i) for i in range(n)]
    an = tu.ResultsAnalyzer(names, pred_px, gt_px, iou_thresh=[0.5])
    an.compute_aggregated_metric("video_mean", np.mean)
    an.compute_aggregated_metric("video_mean", np.mean, metric="map")
    kept = list(an.get_videos_names())
    vm = np.array([an.videos_metrics["video_mean_iou"][k] for k in kept])
    vmap = np.array([an.videos_metrics["video_mean_map_0.5"][k] for k in kept])
    np.savez_compressed(os.path.join(OUT, "opnet_trained_eval.npz"), first=first, n=n, y=y, pred_px=pred_px,
                        kept=np.array([int(k) for k in kept]), video_mean_iou=vm, video_map50=vmap)
    print(f"trained eval: reference mean-IoU {vm.mean():.4f}  mAP@0.5 {vmap.mean():.4f}  ({len(kept)}/{n} videos kept)")


def gen_detector_filter():
    """CaterObjectDetector.remove_low_probability_object (detector.py:14-28) on synthetic detector outputs
    (SURVEY.md 8-c3-vi), incl. an unsorted score list where the prefix rule keeps a low score."""
    from baselines.detector import CaterObjectDetector
    cases = []
    for seed, sort_desc in ((0, True), (1, True), (2, False)):
        rng = np.random.default_rng(seed)
        n = int(rng.integers(3, 12))
        scores = rng.random(n).astype(np.float32)
        if sort_desc:
            scores = np.sort(scores)[::-1].copy()
        boxes = (rng.random((n, 4)) * 300).astype(np.float32)
        labels = rng.integers(0, 193, size=n).astype(np.int64)
        out = CaterObjectDetector.remove_low_probability_object(
            {"boxes": torch.from_numpy(boxes), "labels": torch.from_numpy(labels), "scores": torch.from_numpy(scores)})
        cases.append({"boxes": boxes.tolist(), "labels": labels.tolist(), "scores": scores.tolist(),
                      "kept": int(out["scores"].shape[0]), "kept_boxes_int": out["boxes"].numpy().astype(int).tolist()})
    with open(os.path.join(OUT, "detector_filter.json"), "w") as f:
        json.dump(cases, f)
    print("detector_filter:", [c["kept"] for c in cases])


def gen_analysis():
    """analysis CSV golden: the reference's analyze_results (analyze_iou_offline.py:12-51) on synthetic files"""
    import tempfile
    from baselines.analyze_iou_offline import analyze_results
    with tempfile.TemporaryDirectory() as tmp:
        kw = synth.make_analysis_fixture(tmp)
        out = os.path.join(tmp, "results.csv")
        analyze_results(output_file=out, **kw)
        text = open(out).read()
    with open(os.path.join(OUT, "analysis_results.csv"), "w") as f:
        f.write(text)
    print("analysis csv:", len(text.splitlines()), "lines,", len(text.splitlines()[0].split(",")), "columns")


def sample_indices(name, n, k=4096):
    """deterministic sample of flat indices of a tensor (same helper used by the tests)"""
    if n <= k:
        return np.arange(n)
    u = synth.counter_uniform(synth.name_seed(name, 99), k)
    return np.unique((u * n).astype(np.int64))


def gen_train(lm, cfg, n_clips, t_frames, tag, full, adam_steps):
    """One training step (and a few Adam steps) of the reference model under torch autograd:
    training_main.py:150-152 (Adam lr 1e-3, L1Loss(reduction='none')), :183-217."""
    params = synth.opnet_synth_params(cfg)
    model = lm.OPNet(cfg)
    _load_params(model, params)
    model.train(True)
    boxes, labels = synth.make_batch(0, n_clips, t_frames)
    xb, lb = torch.from_numpy(boxes), torch.from_numpy(labels)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    loss_fn = torch.nn.L1Loss(reduction="none")
    out = {"cfg": np.array(json.dumps(cfg)), "n_clips": n_clips, "t_frames": t_frames}
    losses = []
    for step in range(adam_steps):
        opt.zero_grad()
        y, _ = model(xb)
        loss = torch.mean(loss_fn(y, lb))
        loss.backward()
        losses.append(float(loss.item()))
        if step == 0:
            for k, v in model.named_parameters():
                g = v.grad.detach().numpy()
                out["gnorm/" + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
                if full:
                    out["grad/" + k] = g.copy()
                else:
                    idx = sample_indices(k, g.size)
                    out["gidx/" + k] = idx
                    out["gval/" + k] = g.reshape(-1)[idx].copy()
        opt.step()
    out["losses"] = np.array(losses)
    for k, v in model.state_dict().items():
        w = v.detach().numpy()
        if full:
            out["w_after/" + k] = w.copy()
        else:
            idx = sample_indices(k, w.size)
            out["w_after_val/" + k] = w.reshape(-1)[idx].copy()
    np.savez_compressed(os.path.join(OUT, f"opnet_train_{tag}.npz"), **out)
    print(f"opnet_train_{tag}: losses {losses}")


def gen_sibling_train(lm):
    """one training step (L1 mean, torch autograd) of the reference's BaselineLstm / NonLinearLstm: loss + gradients"""
    out = {}
    cases = [("baseline_lstm", lm.BaselineLstm, synth.baseline_lstm_synth_params, "tiny", {"videos_hidden_dim": 32}, 3, 10),
             ("baseline_lstm", lm.BaselineLstm, synth.baseline_lstm_synth_params, "real", {"videos_hidden_dim": 512}, 3, 120),
             ("non_linear_lstm", lm.NonLinearLstm, synth.non_linear_lstm_synth_params, "tiny",
              {"boxes_features_dim": 16, "videos_hidden_dim": 32}, 3, 10),
             ("non_linear_lstm", lm.NonLinearLstm, synth.non_linear_lstm_synth_params, "real",
              {"boxes_features_dim": 256, "videos_hidden_dim": 512}, 2, 40),
             ("opnet_lstm_mlp", lm.OPNetLstmMlp, synth.opnet_lstm_mlp_synth_params, "tiny",
              {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 32, "videos_hidden_dim": 32}, 3, 10),
             ("opnet_lstm_mlp", lm.OPNetLstmMlp, synth.opnet_lstm_mlp_synth_params, "real",
              {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}, 3, 60),
             # transformer_lstm with its dropout probabilities set to 0 (torch's masks are not reproducible): the
             # reference class in train mode, only the p attributes of its Dropout / MultiheadAttention modules touched
             ("transformer_lstm", lm.TransformerLstm, synth.transformer_lstm_synth_params, "tiny",
              {"boxes_features_dim": 32, "num_attention_heads": 2, "num_attention_layers": 2, "num_lstm_layers": 2,
               "lstm_hidden_dim": 32}, 2, 10),
             ("transformer_lstm", lm.TransformerLstm, synth.transformer_lstm_synth_params, "real",
              {"boxes_features_dim": 256, "num_attention_heads": 2, "num_attention_layers": 2, "num_lstm_layers": 2,
               "lstm_hidden_dim": 512}, 2, 50)]
    for name, cls, pfn, tag, cfg, n, t in cases:
        model = cls(cfg)
        _load_params(model, pfn(cfg))
        model.train(True)
        if name == "transformer_lstm":
            for mod in model.modules():
                if isinstance(mod, torch.nn.Dropout):
                    mod.p = 0.0
                if isinstance(mod, torch.nn.MultiheadAttention):
                    mod.dropout = 0.0
        boxes, labels = synth.make_batch(0, n, t)
        if name == "opnet_lstm_mlp":
            y, _ = model(torch.from_numpy(boxes))
        else:
            y = model(torch.from_numpy(synth.boxes5(boxes)))
        loss = torch.mean(torch.nn.L1Loss(reduction="none")(y, torch.from_numpy(labels)))
        loss.backward()
        pre = f"{name}/{tag}/"
        out[pre + "cfg"] = np.array(json.dumps(cfg)); out[pre + "shape"] = np.array([n, t]); out[pre + "loss"] = np.float64(loss.item())
        for k, v in model.named_parameters():
            g = v.grad.detach().numpy()
            out[pre + "gnorm/" + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
            idx = sample_indices(k, g.size)
            out[pre + "gval/" + k] = g.reshape(-1)[idx].copy()
        print(f"sibling train {name}/{tag}: loss {loss.item():.5f}")
    np.savez_compressed(os.path.join(OUT, "siblings_train.npz"), **out)


def gen_transformer_dropout(lm):
    """TransformerLstm's TRAIN mode as the reference runs it (training_main.py:167 model.train(); learned_models.py:166-168:
    nn.TransformerEncoderLayer's default dropout 0.1 live at four sites per layer): one training step of the reference's own
    class with the dropout masks it drew RECORDED - the fixture holds the masks of slot 0 (the only slot that reaches the loss),
    the loss and the gradients, so that the HIP encoder, fed the same masks, can be held to them at p = 0.1.
    How the masks are seen: torch.nn.functional.dropout is wrapped (draw bernoulli(1 - p), multiply, record) - every nn.Dropout
    goes through it - and scaled_dot_product_attention, whose dropout is internal in torch 2.x, is replaced for the duration of
    the run by softmax(q k^T / sqrt(d)) -> dropout -> @ v, i.e. torch 1.4's MultiheadAttention arithmetic
    (environment.yml:97), which calls the wrapped dropout."""
    import torch.nn.functional as F
    out = {}
    cases = [("tiny", {"boxes_features_dim": 32, "num_attention_heads": 2, "num_attention_layers": 2, "num_lstm_layers": 2,
                       "lstm_hidden_dim": 32}, 2, 10),
             ("real", {"boxes_features_dim": 256, "num_attention_heads": 2, "num_attention_layers": 2, "num_lstm_layers": 2,
                       "lstm_hidden_dim": 512}, 2, 50),
             ("heads4", {"boxes_features_dim": 256, "num_attention_heads": 4, "num_attention_layers": 2, "num_lstm_layers": 2,
                         "lstm_hidden_dim": 512}, 1, 37)]
    real_dropout, real_sdpa = F.dropout, F.scaled_dot_product_attention
    for tag, cfg, n, t in cases:
        torch.manual_seed(1234)
        model = lm.TransformerLstm(cfg)
        _load_params(model, synth.transformer_lstm_synth_params(cfg))
        model.train(True)
        drawn = []

        def dropout(x, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return x
            keep = torch.bernoulli(torch.full_like(x, 1.0 - p))
            drawn.append(keep.to(torch.bool))
            return x * (keep / (1.0 - p))

        def sdpa(q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False, scale=None, **kw):
            assert attn_mask is None and not is_causal
            w = torch.softmax(q @ k.transpose(-2, -1) / (q.shape[-1] ** 0.5 if scale is None else 1.0 / scale), dim=-1)
            return dropout(w, dropout_p, True) @ v

        F.dropout, F.scaled_dot_product_attention = dropout, sdpa
        try:
            boxes, labels = synth.make_batch(0, n, t)
            y = model(torch.from_numpy(synth.boxes5(boxes)))
            loss = torch.mean(torch.nn.L1Loss(reduction="none")(y, torch.from_numpy(labels)))
            loss.backward()
        finally:
            F.dropout, F.scaled_dot_product_attention = real_dropout, real_sdpa
        S, E, H = n * t, cfg["boxes_features_dim"], cfg["num_attention_heads"]
        assert len(drawn) == 4 * cfg["num_attention_layers"], [tuple(d.shape) for d in drawn]
        pre = f"{tag}/"
        out[pre + "cfg"] = np.array(json.dumps(cfg)); out[pre + "shape"] = np.array([n, t]); out[pre + "loss"] = np.float64(loss.item())
        for li in range(cfg["num_attention_layers"]):
            a, d1, f, d2 = drawn[4 * li:4 * li + 4]
            # attention weights [15 slots, heads, S, S] (the encoder's batch axis is the slot axis), the others [S, 15, *]
            assert tuple(a.shape) == (15, H, S, S) and tuple(d1.shape) == (S, 15, E) and tuple(f.shape) == (S, 15, 2048) and tuple(d2.shape) == (S, 15, E)
            for site, m in enumerate((a[0], d1[:, 0], f[:, 0], d2[:, 0])):
                out[pre + f"mask/{li}/{site}"] = np.packbits(m.numpy().reshape(-1))
                out[pre + f"mask_shape/{li}/{site}"] = np.array(m.shape)
        for k, v in model.named_parameters():
            g = v.grad.detach().numpy()
            out[pre + "gnorm/" + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
            out[pre + "gval/" + k] = g.reshape(-1)[sample_indices(k, g.size)].copy()
        out[pre + "y"] = y.detach().numpy()
        print(f"transformer dropout train {tag}: loss {loss.item():.5f}, kept {float(drawn[2].float().mean()):.4f}")
    np.savez_compressed(os.path.join(OUT, "transformer_dropout_train.npz"), **out)


def gen_metric(tu, y, labels):
    """ResultsAnalyzer goldens on integer boxes (tracking_utils.py:137-159, 251-256, 278-288)."""
    frame_shapes = np.array([320, 240, 320, 240])
    n = y.shape[0]
    # exactly the reference's post-processing expression (inference_main.py:219)
    pred_px = (np.array(list(y.reshape(-1, 4))) * frame_shapes).reshape((n, 300, 4)).astype(np.int32)
    gt_px = (np.array(list(labels.reshape(-1, 4))) * frame_shapes).reshape((n, 300, 4)).astype(np.int32)
    # a second prediction set with non-trivial overlap: ground truth jittered by a few pixels
    rng = np.random.default_rng(7)
    jit_px = gt_px + rng.integers(-12, 13, size=gt_px.shape).astype(np.int32)
    res = {"pred_px": pred_px, "gt_px": gt_px, "jit_px": jit_px}
    for tag, p in (("pred", pred_px), ("jit", jit_px)):
        names = [str(i) for i in range(n)]
        an = tu.ResultsAnalyzer(names, p, gt_px, iou_thresh=[0.5])
        an.compute_aggregated_metric("video_mean", np.mean)
        an.compute_aggregated_metric("video_mean", np.mean, metric="map")
        # the analyzer drops any video whose prediction array contains the value -100
        # ("defected videos", tracking_utils.py:234-235) - record which ones survived
        kept = list(an.get_videos_names())
        res[f"kept_{tag}"] = np.array([int(k) for k in kept])
        res[f"iou_{tag}"] = np.stack([an.iou_results[k] for k in kept])
        res[f"video_mean_iou_{tag}"] = np.array([an.videos_metrics["video_mean_iou"][k] for k in kept])
        res[f"video_map50_{tag}"] = np.array([an.videos_metrics["video_mean_map_0.5"][k] for k in kept])
    np.savez_compressed(os.path.join(OUT, "metric.npz"), **res)
    print("metric: mean IoU jit =", res["video_mean_iou_jit"].mean(), " mAP50 jit =", res["video_map50_jit"].mean())


def gen_no_labels(lm):
    """The *_no_labels training loss (training_main.py:192-210: masked L1 + 0.5 * consistency) and its gradients under the
    reference's OPNet class and torch autograd; the loss lines are executed as the reference writes them."""
    cfg = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 16, "videos_hidden_dim": 32}
    params = synth.opnet_synth_params(cfg)
    model = lm.OPNet(cfg)
    _load_params(model, params)
    model.train(True)
    n_clips, t_frames = 3, 12
    boxes, labels = synth.make_batch(0, n_clips, t_frames)
    rng = np.random.default_rng(77)
    mask = np.repeat(rng.random((n_clips, t_frames, 1)) < 0.6, 4, axis=2)          # datasets.py:545-547: whole frames
    output, _ = model(torch.from_numpy(boxes))
    loss_function = torch.nn.L1Loss(reduction="none")
    pred_loss = loss_function(output, torch.from_numpy(labels))
    next_output_frames = output[:, 1:, :]
    current_output_frames = output[:, :-1, :]
    consistency_loss = torch.mean(torch.norm(next_output_frames - current_output_frames, p=2, dim=-1))
    pred_loss = pred_loss * torch.from_numpy(mask)
    pred_loss = torch.mean(pred_loss)
    loss = pred_loss + 0.5 * consistency_loss
    loss.backward()
    out = {"cfg": np.array(json.dumps(cfg)), "n_clips": n_clips, "t_frames": t_frames, "mask": mask,
           "loss": np.float64(loss.item()), "pred_loss": np.float64(pred_loss.item()),
           "consistency_loss": np.float64(consistency_loss.item())}
    for k, v in model.named_parameters():
        out["grad/" + k] = v.grad.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, "opnet_no_labels_train.npz"), **out)
    print("no_labels: loss", float(loss), "pred", float(pred_loss), "consistency", float(consistency_loss))


def gen_grid_classes():
    """The reference's 6x6-grid classification (baselines/proj_utils.py:37-75) run as written, with cv2's two functions
    supplied by oracle/homography.py (4-point DLT; projective map).  Stores H and the classes of a lattice of image
    points plus projected floor points."""
    import importlib
    from oracle import homography
    cv2 = sys.modules.get("cv2") or types.ModuleType("cv2")
    cv2.findHomography = homography.find_homography_dlt
    cv2.perspectiveTransform = homography.perspective_transform
    sys.modules["cv2"] = cv2
    sys.path.insert(0, REF)
    sys.modules.pop("baselines.proj_utils", None)
    pu = importlib.import_module("baselines.proj_utils")
    g = np.linspace(-1.0, 1.0, 41)
    cx, cy = [a.reshape(-1) for a in np.meshgrid(g, g)]
    rng = np.random.default_rng(5)
    floor = np.concatenate([rng.uniform(-3, 3, size=(400, 2)), np.full((400, 1), pu.Z)], axis=1)
    pim = pu.project_3d_point(floor)
    cx, cy = np.concatenate([cx, pim[:, 0]]), np.concatenate([cy, pim[:, 1]])
    cls = np.array([pu.get_class_prediction(float(a), float(b)) for a, b in zip(cx, cy)], dtype=np.int64)
    np.savez_compressed(os.path.join(OUT, "grid_classes.npz"), H=np.asarray(pu.H), cx=cx, cy=cy, cls=cls)
    print("grid_classes:", len(cls), "points,", len(np.unique(cls)), "distinct classes")


def gen_detector_preprocess():
    """CaterObjectDetector.__call__'s frame preparation (detector.py:74-80: BGR -> RGB, / 256 - not 255 -, float32, HWC ->
    CHW, batch axis) run as written: cv2.cvtColor is supplied as the channel reversal it is for COLOR_BGR2RGB and the
    torchvision model is replaced by an identity that hands back the tensor it was given."""
    cv2 = sys.modules.get("cv2") or types.ModuleType("cv2")
    cv2.COLOR_BGR2RGB = 4
    cv2.cvtColor = lambda frame, code: np.ascontiguousarray(frame[..., ::-1]) if code == 4 else (_ for _ in ()).throw(ValueError(code))
    sys.modules["cv2"] = cv2
    sys.path.insert(0, REF)
    import importlib
    sys.modules.pop("baselines.detector", None)
    det_mod = importlib.import_module("baselines.detector")
    det = det_mod.CaterObjectDetector.__new__(det_mod.CaterObjectDetector)
    det.detector = lambda t: t
    frame = np.random.default_rng(11).integers(0, 256, size=(24, 32, 3), dtype=np.uint8)
    x = det(frame, torch.device("cpu"))
    np.savez_compressed(os.path.join(OUT, "detector_preprocess.npz"), frame=frame, tensor=x.numpy())
    print("detector_preprocess:", tuple(x.shape), x.dtype, float(x.max()))


def gen_cone_ids():
    """tests/golden/cone_ids.json: the cone class ids, class count and snitch id of the reference's 193-name table
    (object_indices.py:1-202), obtained by calling its own is_cone_object on every id."""
    sys.path.insert(0, REF)
    import object_indices as roi
    from baselines import datasets as rd
    n = len(roi.OBJECTS_NAME_TO_IDX)
    cones = [i for i in range(n) if roi.is_cone_object(i)]
    with open(os.path.join(OUT, "cone_ids.json"), "w") as f:
        json.dump({"cone_ids": cones, "num_classes": n, "snitch": int(rd.SNITCH_INDEX)}, f)
    print("cone_ids:", len(cones), "cones of", n)


def main():
    """`python oracle/gen_golden.py` rewrites every fixture; `python oracle/gen_golden.py datasets cone_ids ...`
    only the named sections."""
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    only = set(sys.argv[1:])
    want = lambda k: not only or k in only
    lm, tu = _import_reference()
    tiny = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 16, "videos_hidden_dim": 32}
    with open(os.path.join(REF, "configs", "opnet_model_config.json")) as f:
        real = json.load(f)
    if want("opnet"):
        gen_opnet(lm, tiny, n_clips=2, t_frames=12, tag="tiny", keep_intermediates=True)
        y, labels = gen_opnet(lm, real, n_clips=4, t_frames=300, tag="real", keep_intermediates=False)
        gen_metric(tu, y, labels)
    if want("detector_filter"):
        gen_detector_filter()
    if want("analysis"):
        gen_analysis()
    if want("trained"):
        gen_trained(lm, tu, real)
    if want("siblings"):
        gen_siblings(lm)
        gen_sibling_train(lm)
    if want("transformer_dropout"):
        gen_transformer_dropout(lm)
    if want("datasets"):
        gen_datasets()
    if want("cone_ids"):
        gen_cone_ids()
    if want("no_labels"):
        gen_no_labels(lm)
    if want("grid_classes"):
        gen_grid_classes()
    if want("detector_preprocess"):
        gen_detector_preprocess()
    if want("train"):
        gen_train(lm, tiny, n_clips=3, t_frames=12, tag="tiny", full=True, adam_steps=3)
        gen_train(lm, real, n_clips=4, t_frames=300, tag="real", full=False, adam_steps=2)


if __name__ == "__main__":
    main()
