"""End-to-end: synthetic <video>.pkl/_bb.json files + a .pth checkpoint + the reference's JSON config keys ->
objectpermanence_amd.inference_main.reasoning_inference_main -> <video>_bb.json, compared with the CPU oracle
pipeline (encode -> OPNet -> int32 post-process -> IoU)."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

from oracle import opnet_oracle as oo, synth

pytestmark = pytest.mark.gpu
CFG = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}


def test_inference_driver_end_to_end(tmp_path):
    from objectpermanence_amd.datasets import encode_boxes, load_snitch_labels
    from objectpermanence_amd.inference_main import reasoning_inference_main
    s, l, out = tmp_path / "s", tmp_path / "l", tmp_path / "out"
    s.mkdir(); l.mkdir()
    raws = {}
    for i, v in enumerate(["plain", "dups", "crowded", "nosnitch0", "plain"]):
        name = f"v{i}"
        bb, lab, gt = synth.make_raw_video(10 + i, v)
        raws[name] = (bb, lab)
        pickle.dump({"bb": bb, "labels": lab}, open(s / (name + ".pkl"), "wb"), pickle.HIGHEST_PROTOCOL)
        json.dump(gt, open(l / (name + "_bb.json"), "w"))
    params = synth.opnet_synth_params(CFG)
    torch.save({k: torch.from_numpy(v) for k, v in params.items()}, tmp_path / "opnet.pth")
    json.dump(CFG, open(tmp_path / "model.json", "w"))
    json.dump({"batch_size": 2, "num_workers": 0, "device": "cuda:0", "model_path": str(tmp_path / "opnet.pth"),
               "videos_dir": "unused", "sample_dir": str(s), "labels_dir": str(l)}, open(tmp_path / "infer.json", "w"))
    res = reasoning_inference_main("opnet", str(out), str(tmp_path / "infer.json"), str(tmp_path / "model.json"))
    assert res["video_names"] == sorted(raws)
    # oracle pipeline on the same files
    boxes = np.stack([encode_boxes(*raws[n], 6).astype(np.float32) for n in res["video_names"]])
    labels = np.stack([load_snitch_labels(str(l / (n + "_bb.json"))).astype(np.float32) for n in res["video_names"]])
    y, _ = oo.opnet_forward(boxes, params, np.float32)
    px, gt_px = oo.postprocess_to_pixels(y), oo.postprocess_to_pixels(labels)
    assert res["predictions"].shape == px.shape
    assert (res["predictions"] != px).mean() < 2e-3 and np.abs(res["predictions"] - px).max() <= 1
    for n, p in zip(res["video_names"], res["predictions"]):
        on_disk = json.load(open(out / (n + "_bb.json")))
        assert np.array_equal(np.array(on_disk), p) and len(on_disk) == 300
    miou, map50 = oo.mean_iou_and_map(px, gt_px)
    assert res["mean_iou"] == pytest.approx(miou, abs=1e-3)        # north_star: mean-IoU within 1e-3
    assert res["map_0.5"] == pytest.approx(map50, abs=2e-3)


def test_training_driver_end_to_end(tmp_path):
    """training_main mirror on synthetic files: loss falls, dev mean-IoU rises, the best checkpoint is a plain
    state_dict that the inference driver loads back."""
    from objectpermanence_amd.training_main import training_main
    from objectpermanence_amd.inference_main import reasoning_inference_main
    dirs = {}
    for split, n, first in (("train", 24, 100), ("dev", 4, 100)):      # dev = the first 4 training clips
        s, l = tmp_path / f"{split}_s", tmp_path / f"{split}_l"
        s.mkdir(); l.mkdir()
        lines = []
        for i in range(n):
            name = f"{split}{i:02d}"
            bb, lab, gt = synth.make_raw_video(first + i, "plain")
            pickle.dump({"bb": bb, "labels": lab}, open(s / (name + ".pkl"), "wb"), pickle.HIGHEST_PROTOCOL)
            json.dump(gt, open(l / (name + "_bb.json"), "w"))
            lines.append(name + "\t" + ",".join(str(x) for x in range(10 * i, 10 * i + 25)) + "\n")
        open(tmp_path / f"{split}_mask.txt", "w").writelines(lines)
        dirs[split] = (str(s), str(l), str(tmp_path / f"{split}_mask.txt"))
    cfg = {"batch_size": 8, "inference_batch_size": 400, "num_workers": 0, "num_epochs": 40, "print_step": 100,
           "learning_rate": 0.001, "lr_scheduler_patience": 2, "lr_scheduler_factor": 0.8, "device": "cuda:0",
           "checkpoints_path": str(tmp_path / "ckpt"),
           "train_sample_dir": dirs["train"][0], "train_labels_dir": dirs["train"][1], "train_containment_file": dirs["train"][2],
           "dev_sample_dir": dirs["dev"][0], "dev_labels_dir": dirs["dev"][1], "dev_containment_file": dirs["dev"][2]}
    torch.manual_seed(0)
    res = training_main("opnet", cfg, CFG)
    h = res["history"]
    assert h[-1]["train_loss"] < 0.4 * h[0]["train_loss"]
    assert res["best_dev_iou"] > 0.02 and res["checkpoint"] and os.path.exists(res["checkpoint"])
    assert os.path.basename(res["checkpoint"]).endswith(f"_{round(res['best_dev_iou'], 3)}.pth")
    sd = torch.load(res["checkpoint"])
    assert set(sd) == set(synth.opnet_shapes(CFG))
    json.dump(CFG, open(tmp_path / "model.json", "w"))
    json.dump({"batch_size": 4, "num_workers": 0, "device": "cuda:0", "model_path": res["checkpoint"], "videos_dir": "unused",
               "sample_dir": dirs["dev"][0], "labels_dir": dirs["dev"][1]}, open(tmp_path / "infer.json", "w"))
    out = reasoning_inference_main("opnet", str(tmp_path / "out"), str(tmp_path / "infer.json"), str(tmp_path / "model.json"))
    assert len(out["video_names"]) == 4 and np.isfinite(out["mean_iou"])


def test_perception_to_reasoner_end_to_end(tmp_path):
    """config 4 plumbing: raw frames -> detector (HIP backbone + RPN + RoI heads) -> <video>.pkl -> dataset encoder ->
    OPNet -> <video>_bb.json.  Synthetic detector weights: what is checked is the file formats, the 300-frame rule,
    the 0.8 score cut / int truncation, and that batched passes give the per-frame call's detections."""
    from oracle import detector_oracle as do
    from objectpermanence_amd.detector import CaterObjectDetector
    from objectpermanence_amd.inference_main import reasoning_inference_main
    from objectpermanence_amd.preprocess_perception_main import preprocess_main
    vids, res, lab, out = (tmp_path / d for d in ("videos", "perception", "labels", "out"))
    for d in (vids, res, lab):
        d.mkdir()
    rng = np.random.default_rng(0)
    frames = rng.integers(0, 256, size=(300, 60, 80, 3), dtype=np.uint8)
    np.save(vids / "cater_000.npy", frames)
    np.save(vids / "cater_short.npy", frames[:299])                    # not 300 frames: must not be written
    sd = {k: torch.from_numpy(v) for k, v in {**do.synth_backbone_params(), **do.synth_head_params()}.items()}
    torch.save({"model_state_dict": sd}, tmp_path / "detection_model.pth")
    json.dump({"videos_dir": str(vids), "od_model_weights": str(tmp_path / "detection_model.pth"), "device": "cuda:0"},
              open(tmp_path / "preprocess.json", "w"))
    assert preprocess_main(str(res), str(tmp_path / "preprocess.json"), frames_per_pass=12) == 1
    assert sorted(os.listdir(res)) == ["cater_000.pkl"]
    data = pickle.load(open(res / "cater_000.pkl", "rb"))
    assert set(data) == {"bb", "labels"} and len(data["bb"]) == len(data["labels"]) == 300
    for bb, lb in zip(data["bb"], data["labels"]):
        assert bb.dtype.kind == "i" and lb.dtype.kind == "i" and bb.shape == (len(lb), 4)
        assert len(lb) == 0 or (lb.min() >= 1 and lb.max() <= 192 and bb.min() >= 0 and bb[:, [0, 2]].max() <= 80
                                and bb[:, [1, 3]].max() <= 60)
    assert sum(len(lb) for lb in data["labels"]) > 300                 # the synthetic heads do clear 0.8 regularly
    # frame-by-frame call (the reference's pattern, preprocess_perception_main.py:32-36) on a few frames
    det = CaterObjectDetector(str(tmp_path / "detection_model.pth"))
    det.load_model(torch.device("cuda:0"))
    agree = total = 0
    for t in (0, 11, 12, 150, 299):
        one = det.remove_low_probability_object(det(frames[t], torch.device("cuda:0"))[0])
        bb, lb = one["boxes"].cpu().numpy().astype(int), one["labels"].cpu().numpy().astype(int)
        total += len(lb)
        for b, l in zip(bb, lb):
            cand = data["bb"][t][data["labels"][t] == l]
            agree += bool(len(cand) and np.abs(cand - b).max(axis=1).min() <= 1)
    assert total > 0 and agree >= 0.9 * total
    # ... and on into the reasoner through the dataset encoder
    _, _, gt = synth.make_raw_video(3, "plain")
    json.dump(gt, open(lab / "cater_000_bb.json", "w"))
    params = synth.opnet_synth_params(CFG)
    torch.save({k: torch.from_numpy(v) for k, v in params.items()}, tmp_path / "opnet.pth")
    json.dump(CFG, open(tmp_path / "model.json", "w"))
    json.dump({"batch_size": 1, "num_workers": 0, "device": "cuda:0", "model_path": str(tmp_path / "opnet.pth"),
               "videos_dir": str(vids), "sample_dir": str(res), "labels_dir": str(lab)}, open(tmp_path / "infer.json", "w"))
    r = reasoning_inference_main("opnet", str(out), str(tmp_path / "infer.json"), str(tmp_path / "model.json"))
    assert r["video_names"] == ["cater_000"] and r["predictions"].shape == (1, 300, 4)
    assert len(json.load(open(out / "cater_000_bb.json"))) == 300


def test_cater_setup_inference_grid_classes(tmp_path):
    """cater_setup_inference mirror: last-frame prediction -> 6x6 grid class -> class_pred_results.csv, against the
    oracle pipeline (encode -> OPNet -> int32 last-frame box -> centre -> homography -> class)"""
    import pandas as pd
    from objectpermanence_amd.cater_setup_inference import cater_setup_inference, get_classes_predictions, transform_xyxy_to_w_h
    from objectpermanence_amd.datasets import encode_boxes
    s, l, out = tmp_path / "s", tmp_path / "l", tmp_path / "out"
    s.mkdir(); l.mkdir()
    raws = {}
    for i in range(6):
        name = f"CATER_new_{i:06d}"
        bb, lab, gt = synth.make_raw_video(40 + i, "plain")
        raws[name] = (bb, lab)
        pickle.dump({"bb": bb, "labels": lab}, open(s / (name + ".pkl"), "wb"), pickle.HIGHEST_PROTOCOL)
        json.dump(gt, open(l / (name + "_bb.json"), "w"))
    params = synth.opnet_synth_params(CFG)
    torch.save({k: torch.from_numpy(v) for k, v in params.items()}, tmp_path / "opnet.pth")
    json.dump(CFG, open(tmp_path / "model.json", "w"))
    json.dump({"batch_size": 4, "num_workers": 0, "device": "cuda:0", "model_path": str(tmp_path / "opnet.pth"),
               "sample_dir": str(s), "labels_dir": str(l)}, open(tmp_path / "infer.json", "w"))
    df = cater_setup_inference("opnet", str(out), str(tmp_path / "infer.json"), str(tmp_path / "model.json"))
    on_disk = pd.read_csv(out / "class_pred_results.csv")
    assert list(on_disk.columns) == ["video_names", "class_predictions"] and on_disk.equals(df)
    assert df["video_names"].tolist() == [n + ".avi" for n in sorted(raws)]
    boxes = np.stack([encode_boxes(*raws[n], 6).astype(np.float32) for n in sorted(raws)])
    y, _ = oo.opnet_forward(boxes, params, np.float32)
    px = oo.postprocess_to_pixels(y)[:, -1, :]
    want = get_classes_predictions(transform_xyxy_to_w_h(px))
    assert all(0 <= c < 36 for c in want)
    assert df["class_predictions"].tolist() == want


def test_training_driver_transformer_lstm(tmp_path):
    """the training driver on transformer_lstm (train mode = dropout 0.1 from the counter generator, eval per epoch on
    the flash inference path): the loss falls and a checkpoint is written"""
    from objectpermanence_amd.training_main import training_main
    dirs = {}
    for split, n in (("train", 8), ("dev", 4)):
        s, l = tmp_path / f"{split}_s", tmp_path / f"{split}_l"
        s.mkdir(); l.mkdir()
        lines = []
        for i in range(n):
            name = f"{split}{i:02d}"
            bb, lab, gt = synth.make_raw_video(100 + i, "plain")
            pickle.dump({"bb": bb, "labels": lab}, open(s / (name + ".pkl"), "wb"), pickle.HIGHEST_PROTOCOL)
            json.dump(gt, open(l / (name + "_bb.json"), "w"))
            lines.append(name + "\t" + ",".join(str(x) for x in range(10 * i, 10 * i + 25)) + "\n")
        open(tmp_path / f"{split}_mask.txt", "w").writelines(lines)
        dirs[split] = (str(s), str(l), str(tmp_path / f"{split}_mask.txt"))
    cfg = {"batch_size": 4, "inference_batch_size": 4, "num_workers": 0, "num_epochs": 4, "print_step": 100,
           "learning_rate": 0.0005, "lr_scheduler_patience": 2, "lr_scheduler_factor": 0.8, "device": "cuda:0",
           "checkpoints_path": str(tmp_path / "ckpt"),
           "train_sample_dir": dirs["train"][0], "train_labels_dir": dirs["train"][1], "train_containment_file": dirs["train"][2],
           "dev_sample_dir": dirs["dev"][0], "dev_labels_dir": dirs["dev"][1], "dev_containment_file": dirs["dev"][2]}
    mcfg = {"boxes_features_dim": 64, "num_attention_heads": 2, "num_attention_layers": 2, "num_lstm_layers": 2,
            "lstm_hidden_dim": 64}
    torch.manual_seed(0)
    res = training_main("transformer_lstm", cfg, mcfg)
    losses = [h["train_loss"] for h in res["history"]]
    assert len(losses) == 4 and losses[-1] < 0.85 * losses[0]
    assert res["checkpoint"] and os.path.exists(res["checkpoint"])
