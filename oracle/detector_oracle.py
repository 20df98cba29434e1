"""Build-authored torch (CPU) restatement of the detector's preprocessing + ResNet-50-FPN backbone.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: the reference builds
``torchvision.models.detection.fasterrcnn_resnet50_fpn`` (object_detection/models.py:9, torchvision==0.5.0,
environment.yml:113) - third-party code that is neither under /root/reference nor installed here, and
the fine-tuned weights (configs/preprocess_config.json:3 ``detection_model.pth``) are not shipped; the
reference has no tests for it.  This file restates, from torchvision 0.5.0's published architecture,
what the HIP conv path is checked against:
  * the frame conversion of reference baselines/detector.py:74-80 (BGR->RGB, /256, CHW, batch 1);
  * GeneralizedRCNNTransform: normalise (ImageNet mean/std), bilinear resize (align_corners=False) so the
    short side is min_size (cap max_size), zero-pad to a multiple of 32;
  * ResNet-50 (stride on the 3x3 of each bottleneck, FrozenBatchNorm2d = affine with running stats,
    no eps in 0.5.0) and the FPN (1x1 laterals, nearest top-down, 3x3 outputs, LastLevelMaxPool).
  * the region-proposal network (RPNHead, AnchorGenerator sizes 32..512 x ratios 0.5/1/2 with integer strides
    int(padded / grid), BoxCoder weights (1,1,1,1), per-level top-1000 on the raw objectness, clip to the
    resized image, min size 1e-3, per-level NMS 0.7, best 1000);
  * MultiScaleRoIAlign (levels "0".."3", LevelMapper k_min 2 .. k_max 5 around 224/level 4, 7x7 bins, 2x2
    samples per bin, the legacy non-"aligned" roi_align), TwoMLPHead, FastRCNNPredictor (193 classes);
  * postprocess_detections (BoxCoder weights (10,10,5,5), softmax, clip, drop background, score > 0.05, min
    size 1e-2, per-class NMS 0.5 with the CUDA kernel's strict ">" test, best 100) and the transform's box
    rescale to the original frame.
All of it is restated from torchvision 0.5.0's published behaviour, from memory: UNVERIFIED against the library.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import synth
from synthdata.detector import (LAYERS, PLANES, NUM_CLASSES, backbone_shapes, synth_backbone_params,   # noqa: F401
                                head_shapes, synth_head_params)

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)


def resized_size(h: int, w: int, min_size: int = 800, max_size: int = 1333) -> Tuple[int, int]:
    scale = min(float(min_size) / min(h, w), float(max_size) / max(h, w))
    return int(np.floor(h * scale)), int(np.floor(w * scale))       # F.interpolate(scale_factor=...)


def padded_size(rh: int, rw: int, div: int = 32) -> Tuple[int, int]:
    return (rh + div - 1) // div * div, (rw + div - 1) // div * div


def preprocess(frame_bgr: np.ndarray, min_size: int = 800, max_size: int = 1333) -> torch.Tensor:
    """uint8 [H,W,3] BGR -> float32 [1,3,PH,PW]"""
    rgb = frame_bgr[:, :, ::-1].astype(np.float64) / 256                       # detector.py:75-76
    x = torch.as_tensor(rgb.copy(), dtype=torch.float32).permute(2, 0, 1)       # :79-80
    x = (x - torch.tensor(IMAGENET_MEAN)[:, None, None]) / torch.tensor(IMAGENET_STD)[:, None, None]
    rh, rw = resized_size(x.shape[1], x.shape[2], min_size, max_size)
    x = F.interpolate(x[None], size=(rh, rw), mode="bilinear", align_corners=False)
    ph, pw = padded_size(rh, rw)
    out = x.new_zeros((1, 3, ph, pw))
    out[:, :, :rh, :rw] = x
    return out


def _fbn(x, P, prefix, eps):
    scale = P[prefix + ".weight"] / torch.sqrt(P[prefix + ".running_var"] + eps)
    shift = P[prefix + ".bias"] - P[prefix + ".running_mean"] * scale
    return x * scale[None, :, None, None] + shift[None, :, None, None]


def backbone_fpn_forward(x: torch.Tensor, params: Dict[str, np.ndarray], bn_eps: float = 0.0,
                         dtype=torch.float64) -> "OrderedDict[str, torch.Tensor]":
    """x [N,3,H,W] -> OrderedDict {"0","1","2","3","pool"} of [N,256,h,w] (BackboneWithFPN.forward)."""
    P = {k: torch.as_tensor(v, dtype=dtype) for k, v in params.items()}
    x = x.to(dtype)
    b = "backbone.body."
    x = F.relu(_fbn(F.conv2d(x, P[b + "conv1.weight"], stride=2, padding=3), P, b + "bn1", bn_eps))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    feats = []
    for li, nblocks in enumerate(LAYERS, start=1):
        for blk in range(nblocks):
            p = f"{b}layer{li}.{blk}"
            stride = 2 if (blk == 0 and li > 1) else 1
            idt = x
            out = F.relu(_fbn(F.conv2d(x, P[p + ".conv1.weight"]), P, p + ".bn1", bn_eps))
            out = F.relu(_fbn(F.conv2d(out, P[p + ".conv2.weight"], stride=stride, padding=1), P, p + ".bn2", bn_eps))
            out = _fbn(F.conv2d(out, P[p + ".conv3.weight"]), P, p + ".bn3", bn_eps)
            if blk == 0:
                idt = _fbn(F.conv2d(x, P[p + ".downsample.0.weight"], stride=stride), P, p + ".downsample.1", bn_eps)
            x = F.relu(out + idt)
        feats.append(x)
    f = "backbone.fpn."
    last = F.conv2d(feats[3], P[f + "inner_blocks.3.weight"], P[f + "inner_blocks.3.bias"])
    results = [F.conv2d(last, P[f + "layer_blocks.3.weight"], P[f + "layer_blocks.3.bias"], padding=1)]
    for i in (2, 1, 0):
        lat = F.conv2d(feats[i], P[f + f"inner_blocks.{i}.weight"], P[f + f"inner_blocks.{i}.bias"])
        last = lat + F.interpolate(last, size=lat.shape[-2:], mode="nearest")
        results.insert(0, F.conv2d(last, P[f + f"layer_blocks.{i}.weight"], P[f + f"layer_blocks.{i}.bias"], padding=1))
    results.append(F.max_pool2d(results[-1], 1, 2, 0))          # LastLevelMaxPool
    return OrderedDict(zip(["0", "1", "2", "3", "pool"], results))


# ------------------------------------------------------------------------------------------------
# RPN, RoIAlign, box heads, detection post-processing (numpy fp32 for everything that feeds a discrete
# decision - top-k, NMS, thresholds, level mapping - so the same inputs give the same decisions)
# ------------------------------------------------------------------------------------------------
ANCHOR_SIZES = (32, 64, 128, 256, 512)
ASPECT_RATIOS = (0.5, 1.0, 2.0)
BBOX_XFORM_CLIP = float(np.log(1000.0 / 16))
f32 = np.float32


def base_anchors(size: int) -> np.ndarray:
    """AnchorGenerator.generate_anchors in fp32: [3,4] (x1,y1,x2,y2), rounded half-to-even like torch.round"""
    h_ratios = np.sqrt(np.asarray(ASPECT_RATIOS, dtype=f32))
    w_ratios = f32(1.0) / h_ratios
    ws, hs = w_ratios * f32(size), h_ratios * f32(size)
    return np.rint(np.stack([-ws, -hs, ws, hs], axis=1) / f32(2)).astype(f32)


def level_anchors(size: int, gh: int, gw: int, stride_h: int, stride_w: int) -> np.ndarray:
    """grid_anchors for one level: [gh*gw*3, 4], position-major (y, x), anchor-minor"""
    sx = np.arange(gw, dtype=f32) * f32(stride_w)
    sy = np.arange(gh, dtype=f32) * f32(stride_h)
    yy, xx = np.meshgrid(sy, sx, indexing="ij")
    shifts = np.stack([xx.ravel(), yy.ravel(), xx.ravel(), yy.ravel()], axis=1)
    return (shifts[:, None, :] + base_anchors(size)[None, :, :]).reshape(-1, 4).astype(f32)


def decode_boxes(deltas: np.ndarray, boxes: np.ndarray, weights=(1.0, 1.0, 1.0, 1.0)) -> np.ndarray:
    """BoxCoder.decode_single in fp32. deltas [n, 4k], boxes [n, 4] -> [n, 4k]"""
    deltas, boxes = deltas.astype(f32), boxes.astype(f32)
    w = boxes[:, 2] - boxes[:, 0]
    h = boxes[:, 3] - boxes[:, 1]
    cx = boxes[:, 0] + f32(0.5) * w
    cy = boxes[:, 1] + f32(0.5) * h
    dx = deltas[:, 0::4] / f32(weights[0]); dy = deltas[:, 1::4] / f32(weights[1])
    dw = np.minimum(deltas[:, 2::4] / f32(weights[2]), f32(BBOX_XFORM_CLIP))
    dh = np.minimum(deltas[:, 3::4] / f32(weights[3]), f32(BBOX_XFORM_CLIP))
    pcx = dx * w[:, None] + cx[:, None]
    pcy = dy * h[:, None] + cy[:, None]
    pw = np.exp(dw).astype(f32) * w[:, None]
    ph = np.exp(dh).astype(f32) * h[:, None]
    out = np.empty_like(deltas)
    out[:, 0::4] = pcx - f32(0.5) * pw
    out[:, 1::4] = pcy - f32(0.5) * ph
    out[:, 2::4] = pcx + f32(0.5) * pw
    out[:, 3::4] = pcy + f32(0.5) * ph
    return out


def clip_boxes(boxes: np.ndarray, size_hw) -> np.ndarray:
    b = boxes.copy()
    b[..., 0::2] = np.clip(b[..., 0::2], f32(0), f32(size_hw[1]))
    b[..., 1::2] = np.clip(b[..., 1::2], f32(0), f32(size_hw[0]))
    return b


def nms(boxes: np.ndarray, scores: np.ndarray, thresh: float, groups: np.ndarray = None) -> np.ndarray:
    """greedy NMS in fp32, order = stable sort by descending score, suppress when IoU > thresh (strict, as the
    CUDA kernel of torchvision.ops.nms); `groups`: only boxes of the same group interact (batched_nms).
    Returns kept indices in order of decreasing score."""
    boxes = boxes.astype(f32)
    order = np.argsort(-scores.astype(f32), kind="stable")
    b = boxes[order]
    g = None if groups is None else np.asarray(groups)[order]
    area = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    n = len(order)
    dead = np.zeros(n, dtype=bool)
    keep = []
    for i in range(n):
        if dead[i]:
            continue
        keep.append(order[i])
        if i + 1 == n:
            break
        r = b[i + 1:]
        iw = np.maximum(np.minimum(b[i, 2], r[:, 2]) - np.maximum(b[i, 0], r[:, 0]), f32(0))
        ih = np.maximum(np.minimum(b[i, 3], r[:, 3]) - np.maximum(b[i, 1], r[:, 1]), f32(0))
        inter = iw * ih
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = inter / (area[i] + area[i + 1:] - inter)
        hit = iou > f32(thresh)
        if g is not None:
            hit &= g[i + 1:] == g[i]
        dead[i + 1:] |= hit
    return np.asarray(keep, dtype=np.int64)


def rpn_head_forward(feats: "OrderedDict[str, torch.Tensor]", params, dtype=torch.float64):
    """RPNHead on every level -> list of [h, w, 16] arrays: channels 0..2 objectness of anchors 0..2, 3..14 the
    deltas (anchor-major, then dx,dy,dw,dh), channel 15 zero - the layout of the packed HIP head"""
    P = {k: torch.as_tensor(v, dtype=dtype) for k, v in params.items() if k.startswith("rpn.")}
    outs = []
    for f in feats.values():
        t = F.relu(F.conv2d(f.to(dtype), P["rpn.head.conv.weight"], P["rpn.head.conv.bias"], padding=1))
        cls = F.conv2d(t, P["rpn.head.cls_logits.weight"], P["rpn.head.cls_logits.bias"])
        reg = F.conv2d(t, P["rpn.head.bbox_pred.weight"], P["rpn.head.bbox_pred.bias"])
        o = torch.cat([cls, reg, torch.zeros_like(cls[:, :1])], dim=1)[0].permute(1, 2, 0)
        outs.append(o.to(torch.float32).numpy())
    return outs


def rpn_proposals(head_outs, image_size, padded_size_hw, pre_nms_top_n=1000, post_nms_top_n=1000, nms_thresh=0.7,
                  min_size=1e-3):
    """RegionProposalNetwork.filter_proposals for one image. head_outs: per level [h, w, 16] fp32 (see above).
    Returns (proposals [n,4] fp32, scores [n] fp32, levels [n])"""
    cand_b, cand_s, cand_l = [], [], []
    for lvl, o in enumerate(head_outs):
        gh, gw = o.shape[:2]
        sh, sw = int(padded_size_hw[0] / gh), int(padded_size_hw[1] / gw)
        anchors = level_anchors(ANCHOR_SIZES[lvl], gh, gw, sh, sw)
        obj = o[:, :, 0:3].reshape(-1).astype(f32)
        deltas = o[:, :, 3:15].reshape(-1, 4).astype(f32)
        k = min(pre_nms_top_n, obj.shape[0])
        top = np.argsort(-obj, kind="stable")[:k]
        cand_b.append(decode_boxes(deltas[top], anchors[top]))
        cand_s.append(obj[top])
        cand_l.append(np.full(k, lvl))
    boxes, scores, lvls = np.concatenate(cand_b), np.concatenate(cand_s), np.concatenate(cand_l)
    boxes = clip_boxes(boxes, image_size)
    ok = ((boxes[:, 2] - boxes[:, 0]) >= f32(min_size)) & ((boxes[:, 3] - boxes[:, 1]) >= f32(min_size))
    boxes, scores, lvls = boxes[ok], scores[ok], lvls[ok]
    keep = nms(boxes, scores, nms_thresh, lvls)[:post_nms_top_n]
    return boxes[keep], scores[keep], lvls[keep]


def map_levels(boxes: np.ndarray, k_min=2, k_max=5, canonical_scale=224, canonical_level=4, eps=1e-6) -> np.ndarray:
    """LevelMapper in fp32 -> level index 0..3"""
    b = boxes.astype(f32)
    s = np.sqrt((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]))
    with np.errstate(divide="ignore"):
        t = np.floor(f32(canonical_level) + np.log2(s / f32(canonical_scale) + f32(eps)))
    return (np.clip(t, k_min, k_max) - k_min).astype(np.int64)


def roi_align(feat_hwc: np.ndarray, rois: np.ndarray, scale: float, out: int = 7, sampling: int = 2) -> np.ndarray:
    """legacy roi_align (no half-pixel shift), fp32; feat [H,W,C], rois [n,4] -> [n, out, out, C]"""
    H, W, C = feat_hwc.shape
    n = rois.shape[0]
    if n == 0:
        return np.zeros((0, out, out, C), f32)
    r = rois.astype(f32) * f32(scale)
    x0, y0 = r[:, 0], r[:, 1]
    rw = np.maximum(r[:, 2] - x0, f32(1)); rh = np.maximum(r[:, 3] - y0, f32(1))
    bw, bh = rw / f32(out), rh / f32(out)
    res = np.zeros((n, out, out, C), f32)
    p = np.arange(out, dtype=f32)
    for iy in range(sampling):
        y = y0[:, None] + p[None, :] * bh[:, None] + f32(iy + 0.5) * bh[:, None] / f32(sampling)       # [n, out]
        for ix in range(sampling):
            x = x0[:, None] + p[None, :] * bw[:, None] + f32(ix + 0.5) * bw[:, None] / f32(sampling)   # [n, out]
            yy = np.broadcast_to(y[:, :, None], (n, out, out)); xx = np.broadcast_to(x[:, None, :], (n, out, out))
            empty = (yy < -1.0) | (yy > H) | (xx < -1.0) | (xx > W)
            yc, xc = np.maximum(yy, f32(0)), np.maximum(xx, f32(0))
            yl, xl = yc.astype(np.int64), xc.astype(np.int64)
            ytop, xtop = yl >= H - 1, xl >= W - 1
            yl = np.where(ytop, H - 1, yl); xl = np.where(xtop, W - 1, xl)
            yh = np.where(ytop, H - 1, yl + 1); xh = np.where(xtop, W - 1, xl + 1)
            yc = np.where(ytop, yl.astype(f32), yc); xc = np.where(xtop, xl.astype(f32), xc)
            ly, lx = (yc - yl.astype(f32)).astype(f32), (xc - xl.astype(f32)).astype(f32)
            hy, hx = f32(1) - ly, f32(1) - lx
            v = ((hy * hx)[..., None] * feat_hwc[yl, xl] + (hy * lx)[..., None] * feat_hwc[yl, xh]
                 + (ly * hx)[..., None] * feat_hwc[yh, xl] + (ly * lx)[..., None] * feat_hwc[yh, xh])
            res += np.where(empty[..., None], f32(0), v).astype(f32)
    return res / f32(sampling * sampling)


def multiscale_roi_align(feats_hwc, proposals: np.ndarray, image_size) -> np.ndarray:
    """MultiScaleRoIAlign(["0","1","2","3"], 7, 2): feats_hwc = list of the first four FPN maps [h,w,256] fp32"""
    scales = []
    for f in feats_hwc[:4]:
        approx = float(f.shape[0]) / float(image_size[0])
        scales.append(2.0 ** float(np.round(np.log2(f32(approx)))))
    lv = map_levels(proposals)
    out = np.zeros((proposals.shape[0], 7, 7, feats_hwc[0].shape[2]), f32)
    for l in range(4):
        idx = np.nonzero(lv == l)[0]
        if idx.size:
            out[idx] = roi_align(feats_hwc[l].astype(f32), proposals[idx], scales[l])
    return out


def box_heads_forward(pooled_nhwc: np.ndarray, params, dtype=torch.float64):
    """TwoMLPHead + FastRCNNPredictor; pooled [n,7,7,256] is flattened in torchvision's (C,7,7) order"""
    P = {k: torch.as_tensor(v, dtype=dtype) for k, v in params.items() if k.startswith("roi_heads.")}
    x = torch.as_tensor(pooled_nhwc, dtype=dtype).permute(0, 3, 1, 2).flatten(1)
    x = F.relu(F.linear(x, P["roi_heads.box_head.fc6.weight"], P["roi_heads.box_head.fc6.bias"]))
    x = F.relu(F.linear(x, P["roi_heads.box_head.fc7.weight"], P["roi_heads.box_head.fc7.bias"]))
    cls = F.linear(x, P["roi_heads.box_predictor.cls_score.weight"], P["roi_heads.box_predictor.cls_score.bias"])
    reg = F.linear(x, P["roi_heads.box_predictor.bbox_pred.weight"], P["roi_heads.box_predictor.bbox_pred.bias"])
    return cls.to(torch.float32).numpy(), reg.to(torch.float32).numpy()


def postprocess_detections(class_logits: np.ndarray, box_regression: np.ndarray, proposals: np.ndarray, image_size,
                           original_size, score_thresh=0.05, nms_thresh=0.5, detections_per_img=100):
    """RoIHeads.postprocess_detections + GeneralizedRCNNTransform.postprocess for one image, fp32"""
    n = proposals.shape[0]
    boxes = decode_boxes(box_regression, proposals, (10.0, 10.0, 5.0, 5.0)).reshape(n, -1, 4)
    z = class_logits.astype(f32)
    e = np.exp(z - z.max(axis=1, keepdims=True)).astype(f32)
    scores = (e / e.sum(axis=1, keepdims=True, dtype=f32)).astype(f32)
    boxes = clip_boxes(boxes, image_size)
    labels = np.broadcast_to(np.arange(scores.shape[1])[None, :], scores.shape)
    boxes, scores, labels = boxes[:, 1:].reshape(-1, 4), scores[:, 1:].reshape(-1), labels[:, 1:].reshape(-1)
    ok = scores > f32(score_thresh)
    boxes, scores, labels = boxes[ok], scores[ok], labels[ok]
    ok = ((boxes[:, 2] - boxes[:, 0]) >= f32(1e-2)) & ((boxes[:, 3] - boxes[:, 1]) >= f32(1e-2))
    boxes, scores, labels = boxes[ok], scores[ok], labels[ok]
    keep = nms(boxes, scores, nms_thresh, labels)[:detections_per_img]
    boxes, scores, labels = boxes[keep], scores[keep], labels[keep]
    rh = f32(float(original_size[0]) / float(image_size[0])); rw = f32(float(original_size[1]) / float(image_size[1]))
    boxes = boxes * np.asarray([rw, rh, rw, rh], dtype=f32)
    return {"boxes": boxes.astype(f32), "labels": labels.astype(np.int64), "scores": scores.astype(f32)}


def detector_forward(frame_bgr: np.ndarray, params, min_size=800, max_size=1333, dtype=torch.float64):
    """the whole eval-mode fasterrcnn_resnet50_fpn call of detector.py:84 on one frame; also returns the stages"""
    h, w = frame_bgr.shape[:2]
    image_size = resized_size(h, w, min_size, max_size)
    x = preprocess(frame_bgr, min_size, max_size)
    feats = backbone_fpn_forward(x, params, dtype=dtype)
    head = rpn_head_forward(feats, params, dtype=dtype)
    props, pscores, _ = rpn_proposals(head, image_size, x.shape[-2:])
    fh = [feats[k][0].permute(1, 2, 0).to(torch.float32).numpy() for k in ("0", "1", "2", "3")]
    pooled = multiscale_roi_align(fh, props, image_size)
    cls, reg = box_heads_forward(pooled, params, dtype=dtype)
    det = postprocess_detections(cls, reg, props, image_size, (h, w))
    return det, {"feats": feats, "rpn_head": head, "proposals": props, "proposal_scores": pscores, "pooled": pooled,
                 "class_logits": cls, "box_regression": reg, "image_size": image_size}
