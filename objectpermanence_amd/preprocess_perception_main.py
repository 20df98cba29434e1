"""Perception pre-processing: video frames -> per-frame detections -> `<video>.pkl` {"bb", "labels"}.

Mirror of reference baselines/preprocess_perception_main.py (output_video_predictions :16-45, preprocess_video :48-98,
preprocess_main :101-117): the detector runs on every frame, detections below 0.8 are cut (detector.py:14-28), boxes
and labels are truncated to int (:35-36) and a video is written only if it has exactly 300 frames (:92).

MI355X differences: frames go through the detector `frames_per_pass` at a time (one pass of the dense stages over
the batch) instead of one call per frame (:32), on `config["device"]` / cuda:0 instead of the
hard-coded cuda:2 (:75); the per-frame results are the same lists of arrays.  Video decoding is cv2's in the reference
(tracking_utils.VideoHandling); cv2 is not part of this build, so a video is either an iterable / array of BGR uint8
frames, a `.npy` / `.npz` frame stack, or - when cv2 is importable - any file cv2.VideoCapture opens.
"""
from __future__ import annotations

import json
import pickle
import sys
from pathlib import Path
from typing import Dict, Iterable, List, Sequence, Tuple, Union

import numpy as np
import torch

from .detector import PASSES_IN_FLIGHT, CaterObjectDetector
from .models_factory import ModelsFactory

VIDEO_SUFFIXES = (".avi", ".npy", ".npz")


def get_experiment_videos(config: Dict[str, str]) -> List[str]:
    """reference inference_main.py:22-41 (`*.avi` there; frame stacks `*.npy` / `*.npz` are accepted too)"""
    videos_dir = Path(config["videos_dir"])
    paths = sorted(p for p in videos_dir.iterdir() if p.suffix in VIDEO_SUFFIXES)
    if "sample_file" not in config:
        return [str(p) for p in paths]
    by_name = {p.stem: p for p in paths}
    selected = []
    with open(config["sample_file"], "r") as f:
        for line in f:
            selected.append(str(by_name[Path(line[:-1]).stem]))
    return selected


def read_video_frames(video: Union[str, Path, np.ndarray, Iterable[np.ndarray]]) -> Iterable[np.ndarray]:
    """yields BGR uint8 [H, W, 3] frames"""
    if isinstance(video, (str, Path)):
        path = Path(video)
        if path.suffix == ".npy":
            yield from np.load(str(path), mmap_mode="r")
            return
        if path.suffix == ".npz":
            with np.load(str(path)) as z:
                yield from z[z.files[0]]
            return
        try:
            import cv2
        except ImportError as e:            # pragma: no cover - cv2 is absent from this image
            raise RuntimeError(f"decoding {path.suffix} needs cv2 (reference tracking_utils.VideoHandling); "
                               "pass a frame array or a .npy/.npz frame stack instead") from e
        # reference tracking_utils.VideoHandling (:23-30, :44-45, :56-61): cv2 reports one frame more than the labels cover, so
        # exactly CAP_PROP_FRAME_COUNT - 1 frames are read - a 301-count CATER file yields the 300 frames its labels align with
        # (reading to EOF would yield 301 detections, and preprocess_video only writes a pkl for exactly 300)
        cap = cv2.VideoCapture(str(path))
        if not cap.isOpened():
            raise RuntimeError(f"Unable to open video {path}")
        n_valid = int(cap.get(cv2.CAP_PROP_FRAME_COUNT)) - 1
        try:
            for _ in range(max(n_valid, 0)):
                ok, frame = cap.read()
                if not ok or frame is None:  # (a short file: the reference would hand None to the detector and crash)
                    break
                yield frame
        finally:
            cap.release()
        return
    yield from video


def output_video_predictions(video, detector: CaterObjectDetector, compute_device: torch.device,
                             frames_per_pass: int = 16, accuracy_threshold: float = 0.8
                             ) -> Tuple[List[np.ndarray], List[np.ndarray]]:
    """reference :16-45.  -> (bb_predictions, labels): one int array [n_t, 4] / [n_t] per frame"""
    bb_predictions: List[np.ndarray] = []
    labels: List[np.ndarray] = []
    device = torch.device(compute_device)
    streams = [torch.cuda.Stream(device=device) for _ in range(PASSES_IN_FLIGHT)]
    frames_per_pass = min(frames_per_pass, detector.MAX_FRAMES_PER_PASS)

    def collect(handle):
        for det in handle():
            det = detector.remove_low_probability_object(det, accuracy_threshold)
            bb_predictions.append(det["boxes"].cpu().numpy().astype(int))          # :35 (np.int truncation)
            labels.append(det["labels"].cpu().numpy().astype(int))                 # :36

    # PASSES_IN_FLIGHT passes enqueued at a time on alternating streams: the oldest pass's results are collected once that many
    # are in flight (measured on 16-frame passes: 305 / 324 / 327 / 321 frames/s with 1 / 2 / 3 / 4; one-frame passes 201 / 251 / 276 / 252)
    in_flight: List = []
    n_pass = 0
    pending: List[np.ndarray] = []

    def submit():
        nonlocal n_pass, pending
        with torch.cuda.stream(streams[n_pass % len(streams)]):
            in_flight.append(detector.detect_batch_async(pending, device))
        n_pass += 1
        pending = []
        if len(in_flight) >= len(streams):
            collect(in_flight.pop(0))

    for frame in read_video_frames(video):
        pending.append(np.array(frame, dtype=np.uint8, order="C", copy=True))    # memmapped stacks are read-only
        if len(pending) == frames_per_pass:
            submit()
    if pending:
        submit()
    while in_flight:
        collect(in_flight.pop(0))
    return bb_predictions, labels


def preprocess_video(process_args, device=None, frames_per_pass: int = 16, detector: CaterObjectDetector = None) -> bool:
    """reference :48-98.  process_args = (video_path, od_weights, results_dir); returns whether the pkl was written"""
    video_path, od_weights, results_dir = process_args
    device = torch.device(device if device is not None else "cuda:0")
    if detector is None:
        detector = ModelsFactory.get_detector_model("object_detector", od_weights)
        detector.load_model(device)
    bb_predictions, labels = output_video_predictions(video_path, detector, device, frames_per_pass)
    video_name = Path(video_path).stem
    output_data = {"bb": bb_predictions, "labels": labels}
    if len(output_data["bb"]) == 300 and len(output_data["labels"]) == 300:           # :92
        with open(Path(results_dir) / (video_name + ".pkl"), "wb") as f:
            pickle.dump(output_data, f, pickle.HIGHEST_PROTOCOL)
        return True
    return False


def preprocess_main(results_dir: str, config_path: str, frames_per_pass: int = 16) -> int:
    """reference :101-117; the detector is loaded once (the reference reloads it per video, :79-81).  Videos are
    sharded over the ranks of an initialised torch.distributed job (frames and videos are independent: no collective).
    Returns the number of pkl files this rank wrote."""
    with open(config_path, "rb") as f:
        config = json.load(f)
    videos = get_experiment_videos(config)
    from . import parallel
    device = parallel.resolve_device(config.get("device", "cuda:0"))      # cuda:LOCAL_RANK as one rank of a torchrun job
    rank, world = 0, 1
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
    detector = ModelsFactory.get_detector_model("object_detector", config["od_model_weights"])
    detector.load_model(device)
    written = 0
    for video_path in videos[rank::world]:
        try:                                                                          # :112-117
            written += preprocess_video((video_path, config["od_model_weights"], results_dir), device, frames_per_pass,
                                        detector)
        except Exception as e:          # the reference skips a failing video silently; say which one it was
            print(f"preprocess_main: skipped {video_path}: {type(e).__name__}: {e}", file=sys.stderr)
            continue
    return written
