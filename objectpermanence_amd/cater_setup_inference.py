"""CATER snitch-localisation task: last-frame box -> 6x6 grid class per video -> class_pred_results.csv.

Mirror of reference baselines/cater_setup_inference.py (transform_xyxy_to_w_h :18-20, get_classes_predictions :23-32,
cater_setup_inference :35-105): the reasoner runs on the HIP path, only the LAST frame's prediction is used (:77),
int32 pixel truncation (:91), box centre -> [-1, 1] image coordinates (:28) -> floor-plane homography -> class id.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict, List

import numpy as np
import pandas as pd
import torch

from . import parallel
from .datasets import DatasetsFactory, make_loader
from .inference_main import LOADER_MIN_BATCH
from .launch_monitor import DeferredConsumer, HostEvent, verify_launches
from .serving import ReasonerServer, output_boxes
from .models_factory import ModelsFactory
from .proj_utils import get_class_predictions

W_FRAME = 320
H_FRAME = 240


def transform_xyxy_to_w_h(predictions: np.ndarray) -> np.ndarray:
    p = np.asarray(predictions)
    return np.stack([(p[:, 2] + p[:, 0]) / 2, (p[:, 3] + p[:, 1]) / 2], axis=1)


def get_classes_predictions(predictions: np.ndarray) -> List[int]:
    p = np.asarray(predictions, dtype=np.float64)
    return get_class_predictions(p[:, 0] * 2 / W_FRAME - 1, p[:, 1] * 2 / H_FRAME - 1, nrows=3, ncols=3).tolist()


@parallel.bounded_host_threads
def cater_setup_inference(model_name: str, results_dir: str, inference_config_path: str, model_config_path: str) -> pd.DataFrame:
    with open(inference_config_path, "rb") as f:
        config: Dict[str, str] = json.load(f)
    with open(model_config_path, "rb") as f:
        model_config: Dict[str, int] = json.load(f)
    device = parallel.resolve_device(config["device"])      # cuda:LOCAL_RANK as one rank of a torchrun job
    dataset = DatasetsFactory.get_inference_dataset(model_name, config["sample_dir"], config["labels_dir"])
    # data parallel (not in the reference): this rank's share of the minibatches, last-frame boxes gathered by dataset index
    world, rank, exchange = parallel.world_rank()
    # What the loader hands over per round trip (as reasoning_inference_main): the reference's minibatch for transformer_lstm*
    # (its attention couples the clips of a call); clip-independent reasoners travel at least LOADER_MIN_BATCH together
    batch_size = int(config["batch_size"])
    loader_batch = batch_size if parallel.couples_clips(model_name) else max(batch_size, LOADER_MIN_BATCH)
    batches = parallel.plan_inference_batches(model_name, len(dataset), loader_batch, world, rank)
    num_workers = int(config["num_workers"])
    pin = device.type == "cuda" and num_workers > 0
    loader = make_loader(dataset, batches, device, num_workers, pin_memory=pin)
    model = ModelsFactory.get_model(model_name, model_config, config["model_path"])
    model.eval()
    model.to(device)
    # every minibatch is a request to the server (clip-independent reasoners: concatenated into one persistent launch;
    # transformer_lstm*: merged as segments); of each output only the LAST frame is kept (:77) - 16 B per clip - as soon as its
    # forward is seen complete and clean; one host copy at the end instead of a .cpu() per minibatch
    # evaluation is deterministic per call in the reference: a request's result must not depend on what else shares its pass.  The
    # throughput form of the segmented models (transformer_lstm*: large-tile GEMMs, 16-clip LSTM groups - results equal to rounding
    # only) is an explicit choice: "exact_serving": false in the inference config
    server = ReasonerServer(model, model_name, exact=bool(config.get("exact_serving", True)))
    names: List[str] = []
    last: List[torch.Tensor] = []

    def consume(handle):
        last.append(output_boxes(model_name, handle.result())[:, -1, :].reshape(-1, 4).clone())          # :77

    deferred = DeferredConsumer(model, consume, max_pending=64)
    waiting = []

    def hand_over():
        while waiting and waiting[0].done():
            handle = waiting.pop(0)
            ready = handle._event
            if ready is None:
                ready = torch.cuda.Event() if device.type == "cuda" else HostEvent()
                ready.record()
            deferred.add(ready, handle)

    with torch.no_grad():
        for (boxes, _index_to_track), _y, video_names in loader:
            names.extend(video_names)
            waiting.append(server.submit(boxes.to(device, non_blocking=pin)))
            hand_over()
        server.flush()
        hand_over()
        deferred.drain(block=True, all_=True)
        verify_launches(model)
    last = [t.cpu().numpy() for t in ([torch.cat(last)] if last else [])]
    local = np.concatenate(last) if last else np.zeros((0, 4), dtype=np.float32)
    if exchange:
        index = torch.tensor([i for b in batches for i in b], dtype=torch.int64, device=device)
        local = parallel.all_gather_by_index(torch.from_numpy(local).to(device), index, len(dataset)).cpu().numpy()
        names = list(dataset.videos_names)          # dataset order = global index order
    frame_shapes = np.array([320, 240, 320, 240])
    px = (local * frame_shapes).reshape((len(dataset), 4)).astype(np.int32)   # :91
    classes = get_classes_predictions(transform_xyxy_to_w_h(px))
    results = pd.DataFrame({"video_names": [f"{n}.avi" for n in names], "class_predictions": classes})
    if rank == 0:
        Path(results_dir).mkdir(parents=True, exist_ok=True)
        results.to_csv(f"{results_dir}/class_pred_results.csv", index=False)
    return results
