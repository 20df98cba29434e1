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
