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
