"""Offline analysis - mirror of reference baselines/analyze_iou_offline.py:12-51 and the parts of
ResultsAnalyzer it drives (baselines/tracking_utils.py:161-204, 258-397): per-video overall / masked mean
IoU and mAP@thresholds from `<video>_bb.json` prediction and label directories plus the containment /
visibility TSVs, written as the same CSV (same columns, same order, rounded to 3 decimals).

The reference walks videos and frames in Python dict/list loops; here all videos are stacked into one
[N, T, 4] integer array and every metric is a masked reduction over it.  Output is string-identical to the
reference's CSV on the synthetic fixtures (tests/test_analysis.py).
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np
import pandas as pd

SNITCH_NAME = "small_gold_spl_metal_Spl_0"


def _iou(pred: np.ndarray, gt: np.ndarray) -> np.ndarray:
    """tracking_utils.py:137-159 on [N, T, 4] integer boxes -> [N, T] float64 (inclusive +1 pixel convention)."""
    xa = np.maximum(pred[..., 0], gt[..., 0]); ya = np.maximum(pred[..., 1], gt[..., 1])
    xb = np.minimum(pred[..., 2], gt[..., 2]); yb = np.minimum(pred[..., 3], gt[..., 3])
    inter = np.maximum(xb - xa + 1, 0) * np.maximum(yb - ya + 1, 0)
    a1 = (pred[..., 2] - pred[..., 0] + 1) * (pred[..., 3] - pred[..., 1] + 1)
    a2 = (gt[..., 2] - gt[..., 0] + 1) * (gt[..., 3] - gt[..., 1] + 1)
    with np.errstate(divide="ignore", invalid="ignore"):
        return inter / (a1 + a2 - inter)


class ResultsAnalyzer(object):
    """Vectorised counterpart of tracking_utils.ResultsAnalyzer for equal-length videos."""

    def __init__(self, names: List[str], pred: np.ndarray, gt: np.ndarray, iou_thresh: Optional[List[float]] = None):
        self.videos_names = list(names)
        self.pred, self.gt = np.asarray(pred), np.asarray(gt)
        self.iou = _iou(self.pred, self.gt)
        self.iou_thresh = iou_thresh or []
        self.videos_metrics: Dict[str, np.ndarray] = {}          # column name -> [N] values, insertion-ordered

    @classmethod
    def init_from_files(cls, bb_prediction_dir: str, bb_gt_dir: str, iou_thresh: List[float] = None):
        """tracking_utils.py:161-204: `<name>_bb.json` in both directories, matched by name, sorted by name."""
        preds = {f.stem[:-3]: json.load(open(f, "rb")) for f in Path(bb_prediction_dir).glob("*.json")}
        gts = {}
        for f in Path(bb_gt_dir).glob("*.json"):
            name = f.stem[:-3]
            if name in preds:
                boxes = json.load(open(f, "rb"))[SNITCH_NAME]
                gts[name] = [[x, y, x + w, y + h] for x, y, w, h in boxes]
        names = sorted(preds)
        assert names == sorted(gts), "every prediction file needs a label file"
        return cls(names, np.array([preds[n] for n in names], dtype=np.int64),
                   np.array([gts[n] for n in names], dtype=np.int64), iou_thresh)

    def get_frames_mask(self, frames_file: str) -> np.ndarray:
        """tracking_utils.py:258-276 -> bool [N, T]; a video without a line raises like the reference's dict lookup."""
        idx = {n: i for i, n in enumerate(self.videos_names)}
        mask = np.zeros(self.iou.shape, dtype=bool)
        seen = np.zeros(len(self.videos_names), dtype=bool)
        with open(frames_file, "r") as f:
            for line in f:
                name, frames = line[:-1].split("\t")
                if name not in idx:
                    continue
                seen[idx[name]] = True
                if frames != "":
                    mask[idx[name], np.array(frames.split(","), dtype=np.int64)] = True
        if not seen.all():
            raise KeyError(self.videos_names[int(np.flatnonzero(~seen)[0])])
        return mask

    def _values(self, metric: str):
        if metric == "iou":
            return [("", self.iou)]
        if metric == "map":
            return [(f"_{t}", (self.iou > t)) for t in self.iou_thresh]        # strict >, tracking_utils.py:256
        raise NotImplementedError("This metric is not supported")

    def compute_aggregated_metric(self, aggregations_name: str, metric: str = "iou") -> None:
        """mean over all frames (np.mean for iou, sum/len for map - the same number)"""
        for suffix, v in self._values(metric):
            self.videos_metrics[f"{aggregations_name}_{metric}{suffix}"] = v.mean(axis=1)

    def compute_aggregated_metric_masking_frames(self, aggregation_name: str, mask: np.ndarray, metric: str = "iou") -> None:
        """tracking_utils.py:305-358: mean over the masked frames (NaN when the mask is empty) + the mask ratio."""
        cnt = mask.sum(axis=1)
        for suffix, v in self._values(metric):
            # per video: np.mean (iou) / sum/len (map) over the masked frames, the reference's own summation order
            if metric == "iou":
                val = [float(np.mean(v[i][mask[i]])) if cnt[i] > 0 else np.nan for i in range(len(cnt))]
            else:
                val = [float(v[i][mask[i]].sum() / cnt[i]) if cnt[i] > 0 else np.nan for i in range(len(cnt))]
            self.videos_metrics[f"{aggregation_name}_mean_{metric}{suffix}"] = np.array(val)
        if metric == "iou":
            self.videos_metrics[f"{aggregation_name}_ratio"] = np.where(cnt > 0, cnt / mask.shape[1], 0.0)

    def get_analysis_df(self) -> pd.DataFrame:
        """tracking_utils.py:379-390: names and every column sorted by video name (string order)"""
        order = np.argsort(np.array(self.videos_names, dtype=object), kind="stable")
        data = {"videos_names": [self.videos_names[i] for i in order]}
        for k, v in self.videos_metrics.items():
            data[k] = [float(v[i]) for i in order]
        return pd.DataFrame.from_dict(data)

    def write_results(self, results_filepath: str) -> None:
        self.get_analysis_df().round(3).to_csv(results_filepath, index=None)


def analyze_results(predictions_dir: str, labels_dir: str, output_file: str, containment_annotations: str,
                    containment_only_static: str, containment_with_movements: str, visibility_gt_0: str,
                    visibility_gt_30: str, visibility_gt_99: str, iou_thresh: List[float]):
    """analyze_iou_offline.py:12-51, same argument list and same CSV."""
    an = ResultsAnalyzer.init_from_files(predictions_dir, labels_dir, iou_thresh)
    for metric in ("iou", "map"):
        an.compute_aggregated_metric("overall", metric)
        contained = None
        if containment_annotations is not None:
            contained = an.get_frames_mask(containment_annotations)
            an.compute_aggregated_metric_masking_frames("contained", contained, metric)
        if containment_only_static is not None:
            an.compute_aggregated_metric_masking_frames("static_contained", an.get_frames_mask(containment_only_static), metric)
        if containment_with_movements is not None:
            an.compute_aggregated_metric_masking_frames("contained_with_move", an.get_frames_mask(containment_with_movements), metric)
        if visibility_gt_0 is not None:
            vis0 = an.get_frames_mask(visibility_gt_0)
            an.compute_aggregated_metric_masking_frames("visibility_gt_0", vis0, metric)
            if contained is not None:
                # full occlusion == not visible at all and not contained (analyze_iou_offline.py:35-40)
                an.compute_aggregated_metric_masking_frames("full_occlusion", np.logical_and(~vis0, ~contained), metric)
        if visibility_gt_30 is not None:
            an.compute_aggregated_metric_masking_frames("visibility_gt_30", an.get_frames_mask(visibility_gt_30), metric)
        if visibility_gt_99 is not None:
            an.compute_aggregated_metric_masking_frames("visibility_gt_99", an.get_frames_mask(visibility_gt_99), metric)
    an.write_results(output_file)
    return an
