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
RPN, RoIAlign and the box heads are not restated (and not built).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F

from . import synth

IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)
LAYERS = (3, 4, 6, 3)
PLANES = (64, 128, 256, 512)


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


def backbone_shapes() -> "OrderedDict[str, tuple]":
    """state_dict names/shapes of model.backbone (torchvision naming: backbone.body.*, backbone.fpn.*)."""
    sd = OrderedDict()

    def bn(prefix, c):
        for n in ("weight", "bias", "running_mean", "running_var"):
            sd[f"{prefix}.{n}"] = (c,)

    sd["backbone.body.conv1.weight"] = (64, 3, 7, 7)
    bn("backbone.body.bn1", 64)
    inplanes = 64
    for li, (nblocks, planes) in enumerate(zip(LAYERS, PLANES), start=1):
        for b in range(nblocks):
            p = f"backbone.body.layer{li}.{b}"
            sd[f"{p}.conv1.weight"] = (planes, inplanes, 1, 1); bn(f"{p}.bn1", planes)
            sd[f"{p}.conv2.weight"] = (planes, planes, 3, 3); bn(f"{p}.bn2", planes)
            sd[f"{p}.conv3.weight"] = (planes * 4, planes, 1, 1); bn(f"{p}.bn3", planes * 4)
            if b == 0:
                sd[f"{p}.downsample.0.weight"] = (planes * 4, inplanes, 1, 1); bn(f"{p}.downsample.1", planes * 4)
            inplanes = planes * 4
    for i, c in enumerate((256, 512, 1024, 2048)):
        sd[f"backbone.fpn.inner_blocks.{i}.weight"] = (256, c, 1, 1)
        sd[f"backbone.fpn.inner_blocks.{i}.bias"] = (256,)
        sd[f"backbone.fpn.layer_blocks.{i}.weight"] = (256, 256, 3, 3)
        sd[f"backbone.fpn.layer_blocks.{i}.bias"] = (256,)
    return sd


def synth_backbone_params(salt: int = 0) -> Dict[str, np.ndarray]:
    """deterministic synthetic weights that keep activations O(1) through 50 layers"""
    out = {}
    for name, shape in backbone_shapes().items():
        if name.endswith("running_var"):
            out[name] = (1.0 + synth.synth_tensor(name, shape, 0.4, salt)).astype(np.float32)
        elif name.endswith("running_mean") or name.endswith(".bias") and len(shape) == 1 and "bn" in name:
            out[name] = synth.synth_tensor(name, shape, 0.1, salt)
        elif len(shape) == 1 and name.endswith(".weight"):          # BN scale
            g = 0.25 if name.endswith("bn3.weight") else 1.0        # damp the residual branch
            out[name] = (g * (1.0 + synth.synth_tensor(name, shape, 0.3, salt))).astype(np.float32)
        elif len(shape) == 1:                                        # FPN conv bias / downsample BN bias
            out[name] = synth.synth_tensor(name, shape, 0.1, salt)
        else:
            fan_in = shape[1] * shape[2] * shape[3]
            out[name] = synth.synth_tensor(name, shape, float(np.sqrt(4.5 / fan_in)), salt)
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
