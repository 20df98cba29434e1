"""Seeded synthetic weights for the detector (torchvision 0.5.0 fasterrcnn_resnet50_fpn state_dict names and shapes,
193 classes as reference object_detection/models.py:9-13).  Data only; shared by bench.py, tests/, tools/ and
oracle/detector_oracle.py."""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict

import numpy as np

from . import opnet as synth

LAYERS = (3, 4, 6, 3)            # ResNet-50 bottleneck blocks per stage
PLANES = (64, 128, 256, 512)
NUM_CLASSES = 193                # reference object_detection/models.py:9-13


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



def head_shapes() -> "OrderedDict[str, tuple]":
    sd = OrderedDict()
    sd["rpn.head.conv.weight"] = (256, 256, 3, 3); sd["rpn.head.conv.bias"] = (256,)
    sd["rpn.head.cls_logits.weight"] = (3, 256, 1, 1); sd["rpn.head.cls_logits.bias"] = (3,)
    sd["rpn.head.bbox_pred.weight"] = (12, 256, 1, 1); sd["rpn.head.bbox_pred.bias"] = (12,)
    sd["roi_heads.box_head.fc6.weight"] = (1024, 12544); sd["roi_heads.box_head.fc6.bias"] = (1024,)
    sd["roi_heads.box_head.fc7.weight"] = (1024, 1024); sd["roi_heads.box_head.fc7.bias"] = (1024,)
    sd["roi_heads.box_predictor.cls_score.weight"] = (NUM_CLASSES, 1024); sd["roi_heads.box_predictor.cls_score.bias"] = (NUM_CLASSES,)
    sd["roi_heads.box_predictor.bbox_pred.weight"] = (4 * NUM_CLASSES, 1024); sd["roi_heads.box_predictor.bbox_pred.bias"] = (4 * NUM_CLASSES,)
    return sd


def synth_head_params(salt: int = 0) -> Dict[str, np.ndarray]:
    """synthetic head weights with gains chosen so that the discrete stages have something to do: objectness
    logits spread over several units, box deltas of a few tenths, class posteriors peaky enough that a good
    number of (roi, class) pairs clear 0.05 and some clear 0.8."""
    gains = {"rpn.head.conv.weight": np.sqrt(6.0 / (256 * 9)), "rpn.head.cls_logits.weight": 0.1,
             "rpn.head.bbox_pred.weight": 0.01, "roi_heads.box_head.fc6.weight": 0.25 * np.sqrt(6.0 / 12544),
             "roi_heads.box_head.fc7.weight": np.sqrt(6.0 / 1024), "roi_heads.box_predictor.cls_score.weight": 0.25,
             "roi_heads.box_predictor.bbox_pred.weight": 0.08}
    out = {}
    for name, shape in head_shapes().items():
        out[name] = synth.synth_tensor(name, shape, float(gains.get(name, 0.1)), salt)
    return out


