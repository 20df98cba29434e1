"""Output post-processing and the IoU / mAP metric on device.

Replaces the numpy post-processing of reference baselines/inference_main.py:219 (float64 multiply by
[320,240,320,240], truncation to int32) and ResultsAnalyzer's per-frame IoU / video means
(baselines/tracking_utils.py:137-159, 251-256, 278-288). Integer results are bit-exact with the
reference; aggregation (means over frames / videos) is done in float64 like numpy.
"""
from __future__ import annotations

from typing import Tuple

import torch

from . import _lib


def postprocess_and_iou(y: torch.Tensor, labels: torch.Tensor = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """y, labels: [N, T, 4] fp32 normalised boxes on a ROCm device.
    Returns (pred_px int32 [N,T,4], gt_px int32 [N,T,4] | None, iou float64 [N,T] | None)."""
    if not y.is_cuda:
        raise RuntimeError("postprocess_and_iou runs on the GPU only (no CPU fallback)")
    lib = _lib.load()
    y = y.contiguous().float()
    N, T = int(y.shape[0]), int(y.shape[1])
    dev = y.device
    pred = torch.empty((N, T, 4), dtype=torch.int32, device=dev)
    gt = iou = None
    lab_ptr = gt_ptr = iou_ptr = None
    if labels is not None:
        labels = labels.to(dev).contiguous().float()
        gt = torch.empty((N, T, 4), dtype=torch.int32, device=dev)
        iou = torch.empty((N, T), dtype=torch.float64, device=dev)
        lab_ptr, gt_ptr, iou_ptr = labels.data_ptr(), gt.data_ptr(), iou.data_ptr()
    with torch.cuda.device(dev):
        rc = lib.opnet_postprocess_iou(y.data_ptr(), lab_ptr, pred.data_ptr(), gt_ptr, iou_ptr, N, T,
                                       torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(rc, "opnet_postprocess_iou")
    return pred, gt, iou


def mean_iou_and_map(iou: torch.Tensor, thr: float = 0.5) -> Tuple[float, float]:
    """Dataset mean-IoU and mAP@thr: per-video mean over frames, then mean over videos
    (training_main.py:105-106); mAP counts frames with IoU strictly greater than thr
    (tracking_utils.py:251-256)."""
    video_mean = iou.mean(dim=1)
    video_map = (iou > thr).to(torch.float64).mean(dim=1)
    return float(video_mean.mean().item()), float(video_map.mean().item())
