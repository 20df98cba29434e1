"""Input encoding of the reasoners: `<video>.pkl` + `<video>_bb.json` -> model tensors.

Mirror of reference baselines/datasets.py (dataset classes, same names and __getitem__ tuples) and
datasets_factory.py.  The reference encodes a clip with nested Python loops and cmp_to_key sorts
(16.7 ms/clip); here the slot assignment is vectorised with numpy:

  slot order   = the video's distinct class ids, snitch (140) first then ascending (datasets.py:47-54,
                 :271-274), truncated to 15 slots (:291-292);
  slot content = the FIRST occurrence (in the frame's original order) of that class id in the frame
                 (a stable sort + "skip repeated id" walk, :294-309, selects exactly that; for a repeated
                 snitch id the reference's inconsistent comparator selects the LAST one instead), as
                 [x1,y1,x2,y2,1(,is_cone)]; a missing object is all zeros, except that a missing cone keeps
                 its cone bit (:311-318); then divide by [320,240,320,240,1(,1)] in float64 and cast to fp32.

Results are bit-identical to the reference's (tests/test_datasets.py vs tests/golden/datasets.npz).
The dataset classes run the NATIVE form of the same two functions (csrc/encode_host.cpp through the C ABI's
opnet_encode_clips_f32: flat detection arrays in, ~15 us a clip); the numpy form below is the readable statement it is held
to bit for bit (tests/test_datasets.py, oracle/fuzz_datasets.py) and the fallback when the library is absent.
The heuristic "object to track" index vector (:199-257, :338-416) is sequential by nature and is restated
as a small state machine; no loss uses it but it is part of the dataset tuple.
"""
from __future__ import annotations

import json
import os
import pickle
from pathlib import Path
from typing import Dict, List, Tuple

import numpy as np
import torch
from torch.utils.data import Dataset

from .object_indices import CONE_IDS, SNITCH_INDEX
from .supported_models import TRAINING_SUPPORTED_MODELS_5_TRACKS, TRAINING_SUPPORTED_MODELS_6_TRACKS

SNITCH_NAME = "small_gold_spl_metal_Spl_0"
VIDEO_NUM_FRAMES = 300
MAX_OBJECTS = 15
FRAME_SHAPES = np.array([320, 240, 320, 240], dtype=np.float64)


def _flat_ids(labels: List[np.ndarray]) -> np.ndarray:
    """all detection ids of a clip in frame order (one concatenate; frames may be arrays, lists or empty)"""
    parts = [np.asarray(l, dtype=np.int64).reshape(-1) for l in labels]
    return np.concatenate(parts) if parts else np.zeros(0, dtype=np.int64)


def slot_order(labels: List[np.ndarray], all_ids: np.ndarray = None) -> List[int]:
    ids = np.unique(_flat_ids(labels) if all_ids is None else all_ids).tolist()
    rest = [i for i in ids if i != SNITCH_INDEX]                      # np.unique sorts
    return ([SNITCH_INDEX] if len(rest) != len(ids) else []) + rest


def encode_boxes(bb: List[np.ndarray], labels: List[np.ndarray], n_tracks: int = 6) -> np.ndarray:
    """-> float64 [T, 15, n_tracks] normalised boxes (cast to float32 by the caller, like the reference)."""
    T = len(labels)
    all_ids = _flat_ids(labels)
    full_order = slot_order(labels, all_ids)
    order = full_order[:MAX_OBJECTS]
    out = np.zeros((T, MAX_OBJECTS, n_tracks), dtype=np.float64)
    counts = np.fromiter((len(l) for l in labels), dtype=np.int64, count=T)
    if len(all_ids) > 0:
        all_bb = np.concatenate([b for b, n in zip(bb, counts) if n > 0], axis=None).astype(np.float64).reshape(-1, 4)
        frame = np.repeat(np.arange(T, dtype=np.int64), counts)
        lut = np.full(max(int(all_ids.max()) + 1, SNITCH_INDEX + 1), -1, dtype=np.int64)
        for s, oid in enumerate(full_order):
            lut[oid] = s
        rank = lut[all_ids]                                       # position in the video's order, truncated ids too
        if n_tracks == 6:
            # An empty cone slot keeps its cone bit ONLY while the reference's walk still has detections to place
            # (datasets.py:288-318: the while loop ends with the frame's last sorted detection, everything behind it -
            # a whole empty frame included - is plain zero padding, :320-324): slot s of frame t gets [0,0,0,0,0,1]
            # iff s is a cone and s < the largest rank present in t.  An id truncated beyond slot 15 has rank >= 15:
            # the walk then pads every remaining slot before it breaks (:291-292).
            last_rank = np.full(T, -1, dtype=np.int64)
            np.maximum.at(last_rank, frame, rank)
            cone = np.array([1.0 if oid in CONE_IDS else 0.0 for oid in order], dtype=np.float64)
            out[:, :len(order), 5] = cone[None, :] * (np.arange(len(order))[None, :] < last_rank[:, None])
        slot = np.where(rank < MAX_OBJECTS, rank, -1)
        keep = slot >= 0
        key = frame[keep] * MAX_OBJECTS + slot[keep]
        _, first = np.unique(key, return_index=True)              # first occurrence of every (frame, slot)
        rows = np.flatnonzero(keep)[first]
        # ... except for a repeated SNITCH id: the reference's comparator returns -1 for (snitch, snitch) in
        # either order (datasets.py:47-54), so Python's insertion sort puts every later snitch in FRONT of the
        # earlier ones and the walk then takes the LAST occurrence of the frame (pinned by the "dups" golden)
        snitch_rows = np.flatnonzero(all_ids == SNITCH_INDEX)
        if len(snitch_rows) > 0 and lut[SNITCH_INDEX] >= 0:
            rev = snitch_rows[::-1]
            _, last = np.unique(frame[rev], return_index=True)
            rows = np.concatenate([rows[slot[rows] != lut[SNITCH_INDEX]], rev[last]])
        out[frame[rows], slot[rows], :4] = all_bb[rows]
        out[frame[rows], slot[rows], 4] = 1.0
        if n_tracks == 6:
            out[frame[rows], slot[rows], 5] = cone[slot[rows]]    # a detected object carries is_cone_object(id) (:302)
    out[..., :4] /= FRAME_SHAPES
    return out


_CONE_TABLE = None


def _cone_table() -> np.ndarray:
    global _CONE_TABLE
    if _CONE_TABLE is None:
        from .object_indices import NUM_CLASSES
        t = np.zeros(NUM_CLASSES, dtype=np.uint8)
        t[sorted(CONE_IDS)] = 1
        _CONE_TABLE = t
    return _CONE_TABLE


def flatten_detections(bb: List[np.ndarray], labels: List[np.ndarray]):
    """one clip's per-frame detection lists -> (counts int32 [T], ids int32 [N], boxes int32 [N, 4]): the flat form the native
    encoder reads"""
    T = len(labels)
    counts = np.fromiter(map(len, labels), dtype=np.int32, count=T)
    n = int(counts.sum())
    if n == 0:
        return counts, np.zeros(0, dtype=np.int32), np.zeros((0, 4), dtype=np.int32)
    try:            # the files the reference writes: every frame an [n] / [n, 4] integer array (preprocess_perception_main.py:35-36)
        ids = np.concatenate(labels)
        boxes = np.concatenate(bb)
        if ids.ndim != 1 or boxes.ndim != 2 or boxes.shape != (n, 4) or ids.shape[0] != n:
            raise ValueError
    except ValueError:   # lists, 0-d / empty frames of another rank: normalise frame by frame
        ids = np.concatenate([np.asarray(l).reshape(-1) for l in labels])
        boxes = np.concatenate([np.asarray(b).reshape(-1, 4) for b, c in zip(bb, counts) if c > 0])
    return counts, np.ascontiguousarray(ids, dtype=np.int32), np.ascontiguousarray(boxes, dtype=np.int32)


def encode_clips_native(counts: np.ndarray, ids: np.ndarray, boxes: np.ndarray, n_clips: int, T: int, n_tracks: int = 6,
                        clip_first_det: np.ndarray = None, with_index: bool = True, out: np.ndarray = None,
                        idx: np.ndarray = None):
    """`n_clips` clips of T frames each as flat arrays (counts [n_clips * T], ids [N], boxes [N, 4]) -> (float32
    [n_clips, T, 15, n_tracks], int64 [n_clips, T] | None) through the native encoder of libopnet_hip.so
    (csrc/encode_host.cpp: host code, no GPU).  Bit-identical to encode_boxes + index_to_track."""
    lib = _encoder_lib()
    counts = np.ascontiguousarray(counts, dtype=np.int32)
    ids = np.ascontiguousarray(ids, dtype=np.int32)
    boxes = np.ascontiguousarray(boxes, dtype=np.int32).reshape(-1, 4)
    if counts.shape[0] != n_clips * T or ids.shape[0] != boxes.shape[0] or int(counts.sum(dtype=np.int64)) != ids.shape[0] \
            or (counts.size and int(counts.min()) < 0):
        raise ValueError("counts must hold n_clips * T frames and ids / boxes one row per counted detection")
    first_frame = np.arange(n_clips, dtype=np.int64) * T
    if clip_first_det is None:
        per_clip = counts.reshape(n_clips, T).sum(axis=1, dtype=np.int64)
        clip_first_det = np.concatenate([[0], np.cumsum(per_clip)]).astype(np.int64)
    clip_first_det = np.ascontiguousarray(clip_first_det, dtype=np.int64)
    if clip_first_det.shape != (n_clips + 1,) or clip_first_det[0] != 0 or clip_first_det[-1] != ids.shape[0] or \
            np.any(np.diff(clip_first_det) < 0):
        raise ValueError("clip_first_det must be n_clips + 1 non-decreasing offsets from 0 to the number of detections")
    if out is None:         # (a steady-state loader passes its own - e.g. pinned - buffers: fresh pages cost more than the encode)
        out = np.empty((n_clips, T, MAX_OBJECTS, n_tracks), dtype=np.float32)
    if idx is None and with_index:
        idx = np.empty((n_clips, T), dtype=np.int64)
    if out.shape != (n_clips, T, MAX_OBJECTS, n_tracks) or out.dtype != np.float32 or not out.flags.c_contiguous:
        raise ValueError("out must be a C-contiguous float32 [n_clips, T, 15, n_tracks] array")
    cone = _cone_table()
    rc = lib.opnet_encode_clips_f32(counts.ctypes.data, ids.ctypes.data, boxes.ctypes.data, first_frame.ctypes.data,
                                    clip_first_det.ctypes.data, n_clips, T, n_tracks, cone.ctypes.data, int(cone.shape[0]),
                                    out.ctypes.data, idx.ctypes.data if with_index else None)
    if rc != 0:
        raise ValueError(f"opnet_encode_clips_f32 failed (code {rc}): inconsistent counts / shapes")
    return out, idx


_ENC_LIB = None


def _encoder_lib():
    """libopnet_encode.so: csrc/encode_host.cpp built alone (host code only).  The same entry point is exported by
    libopnet_hip.so (the C ABI of include/opnet_hip.h); dataset workers load this sibling so that they never start the HIP
    runtime (1.1 s per worker process before its first sample, measured)."""
    global _ENC_LIB
    if _ENC_LIB is None:
        import ctypes
        from . import build as _build
        path = _build.ENCODE_LIB
        if not os.path.exists(path):
            _build.build_encoder()
        lib = ctypes.CDLL(path)
        vp = ctypes.c_void_p
        lib.opnet_encode_clips_f32.restype = ctypes.c_int
        lib.opnet_encode_clips_f32.argtypes = [vp, vp, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp, vp]
        lib.opnet_load_clips_f32.restype = ctypes.c_int
        lib.opnet_load_clips_f32.argtypes = [vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp, ctypes.c_int, vp, vp, vp,
                                             ctypes.c_char_p, ctypes.c_int]
        lib.opnet_load_clips_mt_f32.restype = ctypes.c_int
        lib.opnet_load_clips_mt_f32.argtypes = lib.opnet_load_clips_f32.argtypes + [ctypes.c_int]
        _ENC_LIB = lib
    return _ENC_LIB


class ClipFileError(ValueError):
    """a <video>.pkl / <video>_bb.json that the native reader refuses (anything but the reference's own format) or cannot read"""


_refusal_warned = False


# Refusals that only say "a plain numeric array in a layout the native reader was not built for": the reference's pickle.load reads
# such a file, and handing it to pickle.load is as safe as the file's numeric content.
_BENIGN_REFUSALS = ("big-endian arrays are not read", "only C-contiguous arrays are read")


def refused_by_native_reader(err: "ClipFileError") -> None:
    """The restricted native reader refuses what it was not built for.  pickle.load executes what a file tells it to, and the files the
    safe reader rejects as malformed or unexpected (an opcode outside the subset, a global other than numpy's ndarray / dtype
    reconstruction, a REDUCE of something else, object dtypes ...) are exactly the ones that must NOT reach it - so a refusal is
    final, with two exceptions: the benign layout refusals above (Fortran order, big-endian: plain numeric arrays
    written by another numpy / platform) fall back to pickle.load + encode for that minibatch with one warning, and
    OPNET_NATIVE_FALLBACK=1 opts in to that fallback for EVERY refusal (trusted files in a format this reader does not know).
    OPNET_NATIVE_STRICT=1 makes every refusal final, the benign ones included."""
    global _refusal_warned
    if os.environ.get("OPNET_NATIVE_STRICT", "0") == "1":
        raise err
    benign = any(m in str(err) for m in _BENIGN_REFUSALS)
    if not benign and os.environ.get("OPNET_NATIVE_FALLBACK", "0") != "1":
        raise ClipFileError(f"{err} - the file is NOT handed to pickle.load (set OPNET_NATIVE_FALLBACK=1 to read files you trust "
                            "with pickle.load instead)") from err
    if not _refusal_warned:
        _refusal_warned = True
        import warnings
        warnings.warn(f"objectpermanence_amd: the native clip-file reader refused a file ({err}); reading such minibatches with "
                      "pickle.load instead (OPNET_NATIVE_STRICT=1 makes this an error)", RuntimeWarning, stacklevel=3)


def native_reader_enabled() -> bool:
    """the restricted native reader of the reference's clip files (csrc/clipfile_host.cpp); OPNET_NATIVE_PKL=0 = pickle.load +
    json.load + the encoder on their arrays, as before"""
    return os.environ.get("OPNET_NATIVE_PKL", "1") != "0" and native_encoder_available()


def load_clips_native(pkl_paths: List[str], json_paths: List[str] = None, T: int = VIDEO_NUM_FRAMES, n_tracks: int = 6,
                      with_index: bool = True, threads: int = 1, out=None):
    """files -> (boxes fp32 [n, T, 15, n_tracks], index int64 [n, T] | None, labels fp32 [n, T, 4] | None) in ONE native call:
    the restricted unpickler of csrc/clipfile_host.cpp (exactly what preprocess_perception_main.py:87-96 writes: a dict of
    lists of numeric ndarrays, protocols 2-5; everything else is refused with ClipFileError), the snitch's labels out of the
    `_bb.json` files (datasets.py:33-45) and the input encoder - bit-identical to pickle.load / json.load + encode_boxes.
    threads > 1: the clips are spread over that many host threads inside the call (the GIL is released); out = (boxes, idx,
    labels) numpy views to fill instead of fresh arrays (a loader's pinned buffers)."""
    import ctypes
    lib = _encoder_lib()
    n = len(pkl_paths)
    if out is not None:
        boxes, idx, labels = out
        if boxes.shape != (n, T, MAX_OBJECTS, n_tracks) or boxes.dtype != np.float32 or not boxes.flags.c_contiguous:
            raise ValueError("out[0] must be a C-contiguous float32 [n, T, 15, n_tracks] array")
        if labels is not None:
            labels[...] = 0
    else:
        boxes = np.empty((n, T, MAX_OBJECTS, n_tracks), dtype=np.float32)
        idx = np.empty((n, T), dtype=np.int64) if with_index else None
        labels = np.zeros((n, T, 4), dtype=np.float32) if json_paths is not None else None
    enc = lambda ps: (ctypes.c_char_p * n)(*[os.fsencode(p) if p is not None else None for p in ps])
    cone = _cone_table()
    err = ctypes.create_string_buffer(512)
    rc = lib.opnet_load_clips_mt_f32(enc(pkl_paths), enc(json_paths) if json_paths is not None else None, n, T, n_tracks,
                                     cone.ctypes.data, int(cone.shape[0]), boxes.ctypes.data,
                                     idx.ctypes.data if idx is not None else None,
                                     labels.ctypes.data if (labels is not None and json_paths is not None) else None, err, len(err),
                                     max(1, int(threads)))
    if rc != 0:
        raise ClipFileError(f"opnet_load_clips_f32 failed (code {rc}): {err.value.decode(errors='replace')}")
    return boxes, idx, labels


def native_encoder_available() -> bool:
    if os.environ.get("OPNET_NATIVE_ENCODE", "1") == "0":
        return False
    try:
        return _encoder_lib() is not None
    except Exception:
        return False


def _closest(frame_centers: np.ndarray, last_location: np.ndarray) -> int:
    """slot whose box centre is nearest to the centre of `last_location` (argmin of the Euclidean norm, as
    datasets.py:186-196; the centres of every frame are computed once per clip)"""
    d = frame_centers - np.array([(last_location[0] + last_location[2]) / 2, (last_location[1] + last_location[3]) / 2])
    return int(np.argmin(np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])))


def index_to_track(boxes: np.ndarray) -> List[int]:
    """Heuristic object-to-track vector: datasets.py:199-257 (5 tracks) / :338-416 (6 tracks: only a cone can
    take over the track when the tracked object disappears)."""
    six = boxes.shape[2] == 6
    out: List[int] = []
    stack: List[int] = []
    last = np.zeros(boxes.shape[2])
    cur = 0
    centers = np.stack([(boxes[:, :, 0] + boxes[:, :, 2]) / 2, (boxes[:, :, 1] + boxes[:, :, 3]) / 2], axis=2)
    vis = boxes[:, :, 4] != 0
    cone = boxes[:, :, 5] != 0 if six else None
    for t, fb in enumerate(boxes):
        if vis[t, 0]:
            out.append(0); last = fb[0]; cur = 0; stack = []
        elif cur == 0:
            c = _closest(centers[t], last)
            if six and not fb[c, 5]:
                out.append(0)                                      # occlusion by a non-cone: keep the snitch
            else:
                out.append(c); last = fb[c]; cur = c; stack.append(0)
        elif not fb[cur, 4]:
            c = _closest(centers[t], last)
            if six and not fb[c, 5]:
                out.append(cur)
            else:
                out.append(c); last = fb[c]; stack.append(cur); cur = c
        else:
            prev = stack[-1]
            if fb[prev, 4]:
                stack.pop(-1); out.append(prev); last = fb[prev]; cur = prev
            else:
                out.append(cur); last = fb[cur]
    return out


def load_snitch_labels(path: str) -> np.ndarray:
    """datasets.py:33-45: [x,y,w,h] -> [x,y,x+w,y+h] / [320,240,320,240] (float64)."""
    with open(path, "rb") as f:
        video_labels = json.load(f)
    arr = np.array(video_labels[SNITCH_NAME], dtype=np.int64).reshape(-1, 4)
    xyxy = np.stack([arr[:, 0], arr[:, 1], arr[:, 0] + arr[:, 2], arr[:, 1] + arr[:, 3]], axis=1)
    return xyxy / FRAME_SHAPES


def read_mask_file(path: str, names) -> Dict[str, np.ndarray]:
    """containment / occlusion TSV: `name\\tf1,f2,...\\n`, parsed with line[:-1] like the
    reference (datasets.py:525-534) - a trailing newline is mandatory."""
    names = set(names)
    out: Dict[str, np.ndarray] = {}
    with open(path, "r") as f:
        for line in f:
            line = line[:-1]
            name, frames = line.split(sep="\t")
            if name in names:
                out[name] = np.array([], dtype=np.int64) if len(frames) == 0 else np.array(frames.split(","), dtype=np.int64)
    return out


class CaterAbstractDataset(Dataset):
    n_tracks = 6

    def __init__(self, predictions_dir: str, label_dir: str):
        self.predictions_dir = Path(predictions_dir)
        self.labels_dir = Path(label_dir)
        self.videos_names: List[str] = []
        self.label_paths: Dict[str, str] = {}
        self._native = None          # decided at the first sample (in the process that encodes it - a DataLoader worker)
        self._native_reader = None   # ... and so is the native file reader

    def _init_dataset_if_not_initiated(self) -> None:
        if len(self.videos_names) == 0:
            self.videos_names = sorted(str(p.stem) for p in self.predictions_dir.glob("*.pkl"))   # datasets.py:70-74
            for name in self.videos_names:
                self.label_paths[name] = str(self.labels_dir / (name + "_bb.json"))

    def __len__(self) -> int:
        self._init_dataset_if_not_initiated()
        return len(self.videos_names)

    def _encode_many(self, indices):
        """-> boxes [n, T, 15, F], index vectors [n, T], labels [n, T, 4] (torch, fp32 / int64 / fp32) and the names of dataset
        items `indices`: ONE native call over their files when the native reader is on, sample by sample otherwise"""
        self._init_dataset_if_not_initiated()
        names = [self.videos_names[i] for i in indices]
        if self._native_reader is None:
            self._native_reader = native_reader_enabled()
        if self._native_reader:
            try:
                boxes, idx, labels = load_clips_native([str(self.predictions_dir / (n + ".pkl")) for n in names],
                                                       [self.label_paths[n] for n in names], VIDEO_NUM_FRAMES, self.n_tracks)
                return torch.from_numpy(boxes), torch.from_numpy(idx), torch.from_numpy(labels), names
            except ClipFileError as e:
                refused_by_native_reader(e)          # (re-raises when OPNET_NATIVE_STRICT=1) - this minibatch through pickle.load
            parts = [self._encode_python(i) for i in indices]
        else:
            parts = [self._encode(i) for i in indices]
        return torch.stack([p[0] for p in parts]), torch.stack([p[1] for p in parts]), torch.stack([p[2] for p in parts]), names

    def _encode(self, idx: int):
        self._init_dataset_if_not_initiated()
        if self._native_reader is None:
            self._native_reader = native_reader_enabled()
        if self._native_reader:
            boxes, ivec, labels, names = self._encode_many([idx])
            return boxes[0], ivec[0], labels[0], names[0]
        return self._encode_python(idx)

    def _encode_python(self, idx: int):
        """the reference's own way in (datasets.py:60-64: pickle.load + json.load), then the encoder"""
        self._init_dataset_if_not_initiated()
        name = self.videos_names[idx]
        labels = load_snitch_labels(self.label_paths[name])
        with open(str(self.predictions_dir / (name + ".pkl")), "rb") as f:
            data = pickle.load(f)
        if self._native is None:
            self._native = native_encoder_available()
        if self._native:
            # the native encoder (csrc/encode_host.cpp, ~15 us a clip): flat arrays in, fp32 boxes + index vector out
            counts, ids, flat = flatten_detections(data["bb"], data["labels"])
            boxes, idx = encode_clips_native(counts, ids, flat, 1, len(counts), self.n_tracks)
            return (torch.from_numpy(boxes[0]), torch.from_numpy(idx[0]), torch.tensor(labels, dtype=torch.float32), name)
        boxes = encode_boxes(data["bb"], data["labels"], self.n_tracks)      # the numpy statement of the same function
        idx_vec = index_to_track(boxes)
        return (torch.tensor(boxes, dtype=torch.float32), torch.tensor(idx_vec, dtype=torch.int64),
                torch.tensor(labels, dtype=torch.float32), name)

    def __getitem__(self, idx: int):
        boxes, idx_vec, labels, name = self._encode(idx)
        return (boxes, idx_vec), (labels, torch.tensor([])), name

    def __getitems__(self, indices):
        """a whole minibatch of a DataLoader worker in one native call over its files (torch's fetcher calls this when the
        loader batches: the samples come back as views of one [n, ...] buffer, ready to be collated)"""
        boxes, ivec, labels, names = self._encode_many(list(indices))
        return [((boxes[k], ivec[k]), (labels[k], torch.tensor([])), names[k]) for k in range(len(names))]


class _TrainingMixin:
    def _init_mask(self, mask_annotations_path: str):
        self.mask_annotations_path = mask_annotations_path
        self.mask_frames: Dict[str, np.ndarray] = {}

    def _init_dataset_if_not_initiated(self) -> None:
        if len(self.videos_names) == 0:
            super()._init_dataset_if_not_initiated()
            self.mask_frames = read_mask_file(self.mask_annotations_path, self.videos_names)

    def __getitem__(self, idx: int):
        boxes, idx_vec, labels, name = self._encode(idx)
        mask = np.zeros((VIDEO_NUM_FRAMES, 4), dtype=bool)
        mask[self.mask_frames[name], :] = True                                                   # datasets.py:545-547
        return (boxes, idx_vec), (labels, torch.tensor(mask)), name

    def __getitems__(self, indices):
        boxes, ivec, labels, names = self._encode_many(list(indices))
        out = []
        for k, name in enumerate(names):
            mask = np.zeros((VIDEO_NUM_FRAMES, 4), dtype=bool)
            mask[self.mask_frames[name], :] = True
            out.append(((boxes[k], ivec[k]), (labels[k], torch.tensor(mask)), name))
        return out


class Cater6TracksForObjectsInferenceDataset(CaterAbstractDataset):
    n_tracks = 6


class Cater5TracksForObjectsInferenceDataset(CaterAbstractDataset):
    n_tracks = 5


class Cater6TracksForObjectsTrainingDataset(_TrainingMixin, CaterAbstractDataset):
    n_tracks = 6

    def __init__(self, predictions_dir: str, label_dir: str, mask_annotations_path: str):
        CaterAbstractDataset.__init__(self, predictions_dir, label_dir)
        self._init_mask(mask_annotations_path)


class Cater5TracksForObjectsTrainingDataset(_TrainingMixin, CaterAbstractDataset):
    n_tracks = 5

    def __init__(self, predictions_dir: str, label_dir: str, mask_annotations_path: str):
        CaterAbstractDataset.__init__(self, predictions_dir, label_dir)
        self._init_mask(mask_annotations_path)


class ClipFileLoader:
    """The drivers' loader when the native file reader is on: yields what `DataLoader(dataset, batch_sampler=batches)` yields -
    ((boxes, index), (labels, mask), names) per minibatch - but produced by ONE multi-threaded native call per minibatch
    (opnet_load_clips_mt_f32, `threads` host threads; the JSON config's num_workers) on a prefetch thread, written straight into
    pinned buffers and - given a CUDA device - already on their way to the GPU: boxes / labels / mask are device tensors whose
    copies run on a side stream; the consumer's stream is made to wait for them when the minibatch is handed over.  No worker
    processes, no inter-process queue, no collate copy.  (The reference: torch DataLoader over its Dataset classes,
    baselines/inference_main.py:177, training_main.py:155-159 - kept, and equal tensor for tensor: tests/test_datasets.py.)"""

    def __init__(self, dataset: "CaterAbstractDataset", batches, device=None, threads: int = 8, depth: int = 3):
        dataset._init_dataset_if_not_initiated()
        self.ds, self.batches = dataset, [list(b) for b in batches]
        self.device = torch.device(device) if device is not None else None
        self.cuda = self.device is not None and self.device.type == "cuda"
        self.threads, self.depth = max(1, int(threads)), max(2, int(depth))
        self.with_mask = hasattr(dataset, "mask_frames")

    def __len__(self):
        return len(self.batches)

    def _slot(self, cap: int):
        F = self.ds.n_tracks
        mk = lambda shape, dt: torch.empty(shape, dtype=dt).pin_memory() if self.cuda else torch.empty(shape, dtype=dt)
        return {"boxes": mk((cap, VIDEO_NUM_FRAMES, MAX_OBJECTS, F), torch.float32), "idx": torch.empty((cap, VIDEO_NUM_FRAMES), dtype=torch.int64),
                "labels": mk((cap, VIDEO_NUM_FRAMES, 4), torch.float32), "mask": mk((cap, VIDEO_NUM_FRAMES, 4), torch.bool), "event": None}

    def __iter__(self):
        import queue
        import threading
        if not self.batches:
            return
        cap = max(len(b) for b in self.batches)
        slots = [self._slot(cap) for _ in range(self.depth + 1)]
        q: "queue.Queue" = queue.Queue(maxsize=self.depth - 1)
        stop = threading.Event()
        side = torch.cuda.Stream(device=self.device) if self.cuda else None

        def produce():
            try:
                for k, b in enumerate(self.batches):
                    if stop.is_set():
                        return
                    slot, n = slots[k % len(slots)], len(b)
                    if slot["event"] is not None:
                        slot["event"].synchronize()          # the copies out of this pinned buffer have left it
                    names = [self.ds.videos_names[i] for i in b]
                    try:
                        load_clips_native([str(self.ds.predictions_dir / (m + ".pkl")) for m in names], [self.ds.label_paths[m] for m in names],
                                          VIDEO_NUM_FRAMES, self.ds.n_tracks, threads=self.threads,
                                          out=(slot["boxes"][:n].numpy(), slot["idx"][:n].numpy(), slot["labels"][:n].numpy()))
                    except ClipFileError as e:
                        refused_by_native_reader(e)      # (re-raises when OPNET_NATIVE_STRICT=1) - this minibatch through pickle.load
                        for r, i in enumerate(b):
                            bx, iv, lb, _ = self.ds._encode_python(i)
                            slot["boxes"][r].copy_(bx); slot["idx"][r].copy_(iv); slot["labels"][r].copy_(lb)
                    mask = None
                    if self.with_mask:
                        mview = slot["mask"][:n]
                        mview.zero_()
                        for r, m in enumerate(names):
                            mview[r, torch.from_numpy(np.asarray(self.ds.mask_frames[m], dtype=np.int64))] = True
                        mask = mview
                    idx = slot["idx"][:n].clone()
                    if self.cuda:
                        with torch.cuda.stream(side):
                            boxes = slot["boxes"][:n].to(self.device, non_blocking=True)
                            labels = slot["labels"][:n].to(self.device, non_blocking=True)
                            mask = mask.to(self.device, non_blocking=True) if mask is not None else None
                            ev = torch.cuda.Event()
                            ev.record(side)
                        slot["event"] = ev
                    else:
                        boxes, labels, ev = slot["boxes"][:n].clone(), slot["labels"][:n].clone(), None
                        mask = mask.clone() if mask is not None else None
                    q.put((boxes, idx, labels, mask, names, ev))
                q.put(None)
            except BaseException as e:          # hand the failure (a refused file, say) to the consumer
                q.put(e)

        th = threading.Thread(target=produce, name="opnet-clip-loader", daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                boxes, idx, labels, mask, names, ev = item
                if ev is not None:
                    torch.cuda.current_stream(self.device).wait_event(ev)
                    for t_ in (boxes, labels, mask):
                        if t_ is not None:
                            t_.record_stream(torch.cuda.current_stream(self.device))
                # (an inference dataset's empty per-sample mask collates to [n, 0], inference_main.py:191)
                yield (boxes, idx), (labels, mask if mask is not None else torch.empty((len(names), 0))), names
        finally:
            stop.set()
            while th.is_alive():            # unblock a producer waiting on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    th.join(timeout=0.05)


def make_loader(dataset, batches, device, num_workers: int, pin_memory: bool = False):
    """the minibatch source of the drivers: ClipFileLoader when the native clip-file reader is on (OPNET_NATIVE_PKL / _ENCODE not
    0; OPNET_NATIVE_LOADER=0 keeps torch's DataLoader), else `DataLoader(dataset, batch_sampler=batches, num_workers)`"""
    if native_reader_enabled() and os.environ.get("OPNET_NATIVE_LOADER", "1") != "0":
        return ClipFileLoader(dataset, batches, device, threads=max(1, int(num_workers)))
    from torch.utils import data
    return data.DataLoader(dataset, batch_sampler=batches, num_workers=num_workers, pin_memory=pin_memory)


class DatasetsFactory(object):
    """reference baselines/datasets_factory.py:8-23"""

    @staticmethod
    def get_training_dataset(model_name: str, samples_dir: str, labels_dir: str, mask_file_path: str):
        if model_name in TRAINING_SUPPORTED_MODELS_5_TRACKS:
            return Cater5TracksForObjectsTrainingDataset(samples_dir, labels_dir, mask_file_path)
        elif model_name in TRAINING_SUPPORTED_MODELS_6_TRACKS:
            return Cater6TracksForObjectsTrainingDataset(samples_dir, labels_dir, mask_file_path)

    @staticmethod
    def get_inference_dataset(model_name: str, samples_dir: str, labels_dir: str):
        if model_name in TRAINING_SUPPORTED_MODELS_5_TRACKS:
            return Cater5TracksForObjectsInferenceDataset(samples_dir, labels_dir)
        if model_name in TRAINING_SUPPORTED_MODELS_6_TRACKS:
            return Cater6TracksForObjectsInferenceDataset(samples_dir, labels_dir)
