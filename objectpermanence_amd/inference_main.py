"""Inference driver - mirror of reference baselines/inference_main.py:162-257 (reasoning_inference_main) for
the learned reasoners: JSON configs in, per-video `<video>_bb.json` predictions out.

Same config keys (configs/inference_config.json): batch_size, num_workers, device, model_path, sample_dir,
labels_dir (videos_dir is only used by the reference for its debug AVI, which is not produced here - cv2 /
video IO is outside the hot path).  Differences, all deliberate:
  * the model, the int32 pixel post-process and the IoU metric run in the HIP library;
  * with torch.distributed initialised the videos are sharded in contiguous blocks over the ranks and the
    int32 predictions are all-gathered (parallel.py); every rank returns the full result, rank 0 writes;
  * predictions are written for every dataset video (the reference writes only those it also finds as .avi).
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict, List

import numpy as np
import torch
import torch.distributed as dist
from torch.utils import data

from . import metrics, parallel
from .datasets import DatasetsFactory
from .models_factory import ModelsFactory
from .supported_models import DOUBLE_OUTPUT_MODELS


def write_bb_predictions_to_file(video_name: str, results_dir: str, predictions) -> str:
    """DataHelper.write_bb_predictions_to_file (tracking_utils.py:96-103): list[T] of [x1,y1,x2,y2] ints, indent 2."""
    path = Path(results_dir) / (Path(video_name).stem + "_bb.json")
    rows = [[int(x1), int(y1), int(x2), int(y2)] for [x1, y1, x2, y2] in predictions]
    with open(path, "w") as f:
        json.dump(rows, f, indent=2)
    return str(path)


def reasoning_inference_main(model_name: str, results_dir: str, inference_config_path: str, model_config_path: str,
                             write_files: bool = True) -> Dict[str, object]:
    with open(inference_config_path, "rb") as f:
        config = json.load(f)
    with open(model_config_path, "rb") as f:
        model_config = json.load(f)
    batch_size = int(config["batch_size"])
    num_workers = int(config["num_workers"])
    device = torch.device(config["device"])

    dataset = DatasetsFactory.get_inference_dataset(model_name, config["sample_dir"], config["labels_dir"])
    n_total = len(dataset)
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0
    lo, hi = parallel.shard_range(n_total, world, rank)
    subset = data.Subset(dataset, range(lo, hi))
    loader = data.DataLoader(subset, batch_size=batch_size, num_workers=num_workers)

    model = ModelsFactory.get_model(model_name, model_config, config.get("model_path"))
    model.eval()
    model.to(device)

    names: List[str] = []
    preds, gts, ious = [], [], []
    with torch.no_grad():
        for (boxes, _index_to_track), (labels, _), video_names in loader:
            out = model(boxes.to(device))
            output = out[0] if model_name in DOUBLE_OUTPUT_MODELS else out
            pred_px, gt_px, iou = metrics.postprocess_and_iou(output, labels.to(device))
            preds.append(pred_px); gts.append(gt_px); ious.append(iou)
            names.extend(video_names)
    t_frames = preds[0].shape[1] if preds else 300
    local_pred = torch.cat(preds) if preds else torch.zeros((0, t_frames, 4), dtype=torch.int32, device=device)
    local_iou = torch.cat(ious) if ious else torch.zeros((0, t_frames), dtype=torch.float64, device=device)
    if world > 1:
        all_pred, _ = parallel.all_gather_predictions(local_pred, n_total)
        all_iou, _ = parallel.all_gather_predictions(local_iou, n_total)
        gathered_names: List[List[str]] = [None] * world
        dist.all_gather_object(gathered_names, names)
        names = [n for part in gathered_names for n in part]
    else:
        all_pred, all_iou = local_pred, local_iou
    mean_iou, map50 = metrics.mean_iou_and_map(all_iou, 0.5) if n_total else (float("nan"), float("nan"))
    pred_np = all_pred.cpu().numpy()
    if write_files and rank == 0:
        Path(results_dir).mkdir(parents=True, exist_ok=True)
        for name, p in zip(names, pred_np):
            write_bb_predictions_to_file(name, results_dir, p)
    return {"video_names": names, "predictions": pred_np, "mean_iou": mean_iou, "map_0.5": map50}
