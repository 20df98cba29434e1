"""Inference driver - mirror of reference baselines/inference_main.py:162-257 (reasoning_inference_main) for
the learned reasoners: JSON configs in, per-video `<video>_bb.json` predictions out.

Same config keys (configs/inference_config.json): batch_size, num_workers, device, model_path, sample_dir,
labels_dir (videos_dir is only used by the reference for its debug AVI, which is not produced here - cv2 /
video IO is outside the hot path).  Differences, all deliberate:
  * the model, the int32 pixel post-process and the IoU metric run in the HIP library;
  * with torch.distributed initialised the videos are sharded over the ranks (parallel.plan_inference_batches: contiguous
    blocks for the clip-independent reasoners; WHOLE reference minibatches for transformer_lstm*, whose attention couples
    the clips of a minibatch - a split batch would change its outputs) and the int32 predictions are all-gathered by
    dataset index; every rank returns the full result, rank 0 writes;
  * there is not one forward per DataLoader minibatch: the minibatches are submitted to a serving.ReasonerServer, which runs
    up to 1024 pending clips as one forward (clip-independent reasoners: same outputs, clips are independent) or merges the
    pending minibatches as segments of one pass (transformer_lstm*: each minibatch still attends only to itself);
  * predictions are written for every dataset video (the reference writes only those it also finds as .avi).
"""
from __future__ import annotations

import json
import os
import time
from pathlib import Path
from typing import Dict, List

import numpy as np
import torch
import torch.distributed as dist
from torch.utils import data

from . import metrics, parallel
from .datasets import DatasetsFactory, make_loader
from .launch_monitor import DeferredConsumer, HostEvent, verify_launches
from .models_factory import ModelsFactory
from .serving import ReasonerServer, output_boxes


LOADER_MIN_BATCH = 256     # clips per DataLoader round trip for the clip-independent reasoners (see reasoning_inference_main)


def write_bb_predictions_to_file(video_name: str, results_dir: str, predictions) -> str:
    """DataHelper.write_bb_predictions_to_file (tracking_utils.py:96-103): list[T] of [x1,y1,x2,y2] ints, indent 2."""
    path = Path(results_dir) / (Path(video_name).stem + "_bb.json")
    rows = [[int(x1), int(y1), int(x2), int(y2)] for [x1, y1, x2, y2] in predictions]
    with open(path, "w") as f:
        json.dump(rows, f, indent=2)
    return str(path)


@parallel.bounded_host_threads
def reasoning_inference_main(model_name: str, results_dir: str, inference_config_path: str, model_config_path: str,
                             write_files: bool = True) -> Dict[str, object]:
    with open(inference_config_path, "rb") as f:
        config = json.load(f)
    with open(model_config_path, "rb") as f:
        model_config = json.load(f)
    batch_size = int(config["batch_size"])
    num_workers = int(config["num_workers"])
    # the JSON's device (inference_main.py:189) - or cuda:LOCAL_RANK when this process is one rank of a torchrun job
    device = parallel.resolve_device(config["device"])

    dataset = DatasetsFactory.get_inference_dataset(model_name, config["sample_dir"], config["labels_dir"])
    n_total = len(dataset)
    world, rank, exchange = parallel.world_rank()
    # What the loader hands over per round trip.  transformer_lstm*: exactly the reference's minibatch (its attention couples
    # the clips of a call).  Clip-independent reasoners: any cut of the clips gives the same outputs, and a minibatch of 16
    # (configs/inference_config.json) costs one round trip just like one of 256 - so at least LOADER_MIN_BATCH clips travel
    # together (tools/e2e_inference_time.py: a torch DataLoader with 8 workers and batch 16 delivers 6.3 k clips/s from files on
    # the box, bound by the receiving process's per-batch work; datasets.ClipFileLoader: no processes, no queue, pinned buffers)
    loader_batch = batch_size if parallel.couples_clips(model_name) else max(batch_size, LOADER_MIN_BATCH)
    batches = parallel.plan_inference_batches(model_name, n_total, loader_batch, world, rank)
    pin = device.type == "cuda" and num_workers > 0
    loader = make_loader(dataset, batches, device, num_workers, pin_memory=pin)

    model = ModelsFactory.get_model(model_name, model_config, config.get("model_path"))
    model.eval()
    model.to(device)
    # every DataLoader minibatch is one request.  Clip-independent reasoners: pending requests are concatenated into one
    # forward; transformer_lstm* (attention couples the clips of a minibatch): merged as SEGMENTS - each minibatch still attends
    # only to itself, exactly the reference's per-minibatch call (serving.py)
    # evaluation is deterministic per call in the reference: a request's result must not depend on what else shares its pass.  The
    # throughput form of the segmented models (transformer_lstm*: large-tile GEMMs, 16-clip LSTM groups - results equal to rounding
    # only) is an explicit choice: "exact_serving": false in the inference config
    server = ReasonerServer(model, model_name, exact=bool(config.get("exact_serving", True)))

    names: List[str] = []
    preds, ious = [], []
    t_start, t_first, n_first = time.perf_counter(), None, 0

    def consume(handle, labels_dev):
        """one request's output -> int32 pixel boxes + per-frame IoUs (4.8 + 2.4 KB per clip); the output itself is dropped"""
        output = output_boxes(model_name, handle.result())
        pred_px, _gt_px, iou = metrics.postprocess_and_iou(output, labels_dev)
        preds.append(pred_px); ious.append(iou)

    # A request is post-processed as soon as its forward is seen complete and clean - a persistent launch that gave up (bounded
    # spins, NaN outputs) is re-run on the launch chain into the same output tensors first, so NaN never reaches the int32
    # post-process or the JSON files - instead of holding every output of the data set until one sync at its end
    deferred = DeferredConsumer(model, consume, max_pending=64)
    waiting = []                         # submitted, forward not issued yet (the server is still collecting its pass)

    def hand_over():
        """requests whose forward has been enqueued go to the deferred consumer, in submission order"""
        while waiting and waiting[0][0].done():
            handle, labels_dev = waiting.pop(0)
            ready = handle._event
            if ready is None:
                ready = torch.cuda.Event() if device.type == "cuda" else HostEvent()
                ready.record()
            deferred.add(ready, handle, labels_dev)

    with torch.no_grad():
        for (boxes, _index_to_track), (labels, _), video_names in loader:
            if t_first is None:          # the loader's workers are up and the first minibatch has arrived: steady state from here
                t_first, n_first = time.perf_counter(), len(video_names)
            names.extend(video_names)
            waiting.append((server.submit(boxes.to(device, non_blocking=pin)), labels.to(device, non_blocking=pin)))
            hand_over()
        server.flush()
        hand_over()
        deferred.drain(block=True, all_=True)    # the sync point of this driver
        verify_launches(model)
    t_frames = preds[0].shape[1] if preds else 300
    local_pred = torch.cat(preds) if preds else torch.zeros((0, t_frames, 4), dtype=torch.int32, device=device)
    local_iou = torch.cat(ious) if ious else torch.zeros((0, t_frames), dtype=torch.float64, device=device)
    if exchange:
        index = torch.tensor([i for b in batches for i in b], dtype=torch.int64, device=device)
        all_pred = parallel.all_gather_by_index(local_pred, index, n_total)
        all_iou = parallel.all_gather_by_index(local_iou, index, n_total)
        names = list(dataset.videos_names)          # dataset order = global index order (datasets.py:70-74)
    else:
        all_pred, all_iou = local_pred, local_iou
    mean_iou, map50 = metrics.mean_iou_and_map(all_iou, 0.5) if n_total else (float("nan"), float("nan"))
    pred_np = all_pred.cpu().numpy()
    if write_files and rank == 0:
        Path(results_dir).mkdir(parents=True, exist_ok=True)
        for name, p in zip(names, pred_np):
            write_bb_predictions_to_file(name, results_dir, p)
    t_end = time.perf_counter()
    n_local = sum(len(b) for b in batches)
    timing = {"startup_s": (t_first or t_end) - t_start, "total_s": t_end - t_start,
              # outputs alive at once (requests whose forward had been enqueued but not yet post-processed) and the allocator's peak
              "peak_pending_outputs": deferred.peak_pending,
              "peak_device_bytes": int(torch.cuda.max_memory_allocated(device)) if device.type == "cuda" else None,
              # files -> predictions on the host, this rank's clips, without the DataLoader's worker start-up and first batch
              "steady_clips_per_s": (n_local - n_first) / max(t_end - t_first, 1e-9) if t_first is not None and n_local > n_first else None}
    return {"video_names": names, "predictions": pred_np, "mean_iou": mean_iou, "map_0.5": map50, "timing": timing}
