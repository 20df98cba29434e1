"""Drop-in nn.Modules for the reference's learned reasoners, backed by libopnet_hip.so.

Mirror of reference baselines/learned_models.py: same class names, constructor argument (the JSON
config dict), parameter names/shapes (so reference ``.pth`` state_dicts load unchanged) and
``forward`` signatures/outputs. ``forward`` does not use torch ops for the arithmetic: it hands raw
device pointers to the C ABI (include/opnet_hip.h) on the current HIP stream. There is no CPU
path - inputs must live on a ROCm device and the shared library must be present.
"""
from __future__ import annotations

import math
import os
from typing import Dict, Tuple

import torch
import torch.nn as nn

from . import _lib


class AbstractCaterModel(nn.Module):
    """reference learned_models.py:8-15"""

    def __init__(self, config: Dict[str, int]):
        super().__init__()
        self.config: Dict[str, int] = config
        self.max_objects_in_frame = 15
        self.bb_in_dim = 5
        self.bb_out_dim = 4


class LSTMWeights(nn.Module):
    """Parameter holder with torch.nn.LSTM's parameter names and default init (bias=False,
    unidirectional): weight_ih_l{k} [4H, in], weight_hh_l{k} [4H, H], U(-1/sqrt(H), 1/sqrt(H)).
    It deliberately has no forward: the recurrence runs in the HIP library."""

    def __init__(self, input_size: int, hidden_size: int, num_layers: int = 1):
        super().__init__()
        self.input_size, self.hidden_size, self.num_layers = input_size, hidden_size, num_layers
        bound = 1.0 / math.sqrt(hidden_size)
        for layer in range(num_layers):
            in_dim = input_size if layer == 0 else hidden_size
            for name, shape in ((f"weight_ih_l{layer}", (4 * hidden_size, in_dim)),
                                (f"weight_hh_l{layer}", (4 * hidden_size, hidden_size))):
                p = nn.Parameter(torch.empty(shape))
                nn.init.uniform_(p, -bound, bound)
                self.register_parameter(name, p)


class LinearWeight(nn.Module):
    """Parameter holder with nn.Linear(bias=False)'s name and init (kaiming_uniform(a=sqrt 5) ==
    U(-1/sqrt(in), 1/sqrt(in)))."""

    def __init__(self, in_features: int, out_features: int):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        bound = 1.0 / math.sqrt(in_features)
        nn.init.uniform_(self.weight, -bound, bound)


def _stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class _OPNetTrainFunction(torch.autograd.Function):
    """Autograd bridge: forward = opnet_train_forward_f32 (keeps the history in the module's training
    workspace), backward = opnet_train_backward_f32 (BPTT + weight-gradient GEMMs)."""

    @staticmethod
    def forward(ctx, module, boxes, *weights):
        lib = _lib.load()
        B, T = int(boxes.shape[0]), int(boxes.shape[1])
        dev = boxes.device
        h1, h2 = module._h1, module._h2
        with torch.cuda.device(dev):
            stream = _stream_ptr(dev)
            nbytes = lib.opnet_train_packed_weights_bytes(h1, h2)
            if module._tpacked is None or module._tpacked.device != dev:
                module._tpacked = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
            rc = lib.opnet_train_pack_weights_f32(*(w.data_ptr() for w in weights), module._tpacked.data_ptr(),
                                                  nbytes, h1, h2, stream)
            _lib.check(rc, "opnet_train_pack_weights_f32")
            key = (B, T, str(dev))
            if module._tws_key != key:
                wsb = lib.opnet_train_workspace_bytes(B, T, h1, h2)
                if wsb == 0:
                    _lib.check(-2, "opnet_train_workspace_bytes")
                module._tws = None          # release the old history first
                module._tws = torch.empty(wsb, dtype=torch.uint8, device=dev)
                module._tws_key = key
            y = torch.empty((B, T, 4), dtype=torch.float32, device=dev)
            logits = torch.empty((B, 15, T), dtype=torch.float32, device=dev)
            rc = lib.opnet_train_forward_f32(boxes.data_ptr(), module._tpacked.data_ptr(), y.data_ptr(),
                                             logits.data_ptr(), module._tws.data_ptr(), module._tws.numel(),
                                             B, T, h1, h2, stream)
            _lib.check(rc, "opnet_train_forward_f32")
        module._train_gen += 1
        ctx.module, ctx.gen, ctx.shape = module, module._train_gen, (B, T)
        ctx.wshapes = [tuple(w.shape) for w in weights]
        ctx.mark_non_differentiable(logits)
        ctx.set_materialize_grads(False)
        return y, logits

    @staticmethod
    def backward(ctx, grad_y, grad_logits):
        module = ctx.module
        if ctx.gen != module._train_gen:
            raise RuntimeError("OPNet: backward() after another training forward - the saved history of this "
                               "forward has been overwritten (one history per module)")
        n_in = 2 + len(ctx.wshapes)
        if grad_y is None:
            return (None,) * n_in
        lib = _lib.load()
        B, T = ctx.shape
        dev = grad_y.device
        grad_y = grad_y.contiguous().float()
        grads = [torch.empty(s, dtype=torch.float32, device=dev) for s in ctx.wshapes]
        with torch.cuda.device(dev):
            rc = lib.opnet_train_backward_f32(grad_y.data_ptr(), module._tpacked.data_ptr(), module._tws.data_ptr(),
                                              module._tws.numel(), *(g.data_ptr() for g in grads), B, T,
                                              module._h1, module._h2, _stream_ptr(dev))
        _lib.check(rc, "opnet_train_backward_f32")
        return (None, None) + tuple(grads)


class OPNet(AbstractCaterModel):
    """reference learned_models.py:18-52.  forward(boxes [B,T,15,6]) -> (y_boxes [B,T,4],
    object_to_track_prediction [B,15,T])."""

    def __init__(self, config: Dict[str, int]):
        super().__init__(config)
        self.bb_in_dim = 6
        object_to_track_dim = config["object_to_track_pred_dim"]
        h1 = config["object_to_track_hidden_dim"]
        h2 = config["videos_hidden_dim"]
        if object_to_track_dim != 15:
            # the reference's einsum "bfot,bfo->bft" (:43) only type-checks for 15 slots
            raise ValueError("object_to_track_pred_dim must be 15 (number of object slots)")
        self.object_to_track_LSTM = LSTMWeights(self.bb_in_dim * 15, h1)
        self.object_to_track_prediction = LinearWeight(h1, object_to_track_dim)
        self.video_LSTM = LSTMWeights(self.bb_in_dim, h2)
        self.prediction_layer = LinearWeight(h2, self.bb_out_dim)
        self._h1, self._h2 = h1, h2
        self._packed = None
        self._packed_key = None
        self._plans: Dict[Tuple[int, int, int, int], Tuple[int, torch.Tensor]] = {}
        self._tpacked = None     # training: inference tiles + transposed tiles
        self._tws = None         # training workspace (one forward's history)
        self._tws_key = None
        self._train_gen = 0
        self.use_graph = os.environ.get("OPNET_HIP_EAGER", "0") != "1"

    # -- weights ------------------------------------------------------------------------------
    def _weights(self):
        return (self.object_to_track_LSTM.weight_ih_l0, self.object_to_track_LSTM.weight_hh_l0,
                self.object_to_track_prediction.weight, self.video_LSTM.weight_ih_l0,
                self.video_LSTM.weight_hh_l0, self.prediction_layer.weight)

    def _packed_weights(self, device: torch.device) -> torch.Tensor:
        lib = _lib.load()
        ws = self._weights()
        key = tuple((w.data_ptr(), w._version) for w in ws) + (str(device),)
        if self._packed is None or self._packed_key != key:
            for w in ws:
                if w.device != device or w.dtype != torch.float32 or not w.is_contiguous():
                    raise RuntimeError("OPNet parameters must be contiguous fp32 on the input's device "
                                       "(call model.to(device) first)")
            nbytes = lib.opnet_packed_weights_bytes(self._h1, self._h2)
            if nbytes == 0:
                _lib.check(-2, "opnet_packed_weights_bytes")
            if self._packed is None or self._packed.device != device:
                self._packed = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
            rc = lib.opnet_pack_weights_f32(*(w.data_ptr() for w in ws), self._packed.data_ptr(),
                                            nbytes, self._h1, self._h2, _stream_ptr(device))
            _lib.check(rc, "opnet_pack_weights_f32")
            self._packed_key = key
        return self._packed

    # -- forward ------------------------------------------------------------------------------
    def forward(self, boxes: torch.Tensor):
        if not boxes.is_cuda:
            raise RuntimeError("objectpermanence_amd.OPNet runs on MI355X only: move `boxes` (and the model) "
                               "to a ROCm device; there is no CPU fallback")
        if boxes.dim() != 4 or boxes.shape[2] != 15 or boxes.shape[3] != 6:
            raise ValueError(f"boxes must be [B, T, 15, 6], got {tuple(boxes.shape)}")
        lib = _lib.load()
        boxes = boxes.contiguous().float()
        if torch.is_grad_enabled() and any(w.requires_grad for w in self._weights()):
            ws = self._weights()
            for w in ws:
                if w.device != boxes.device or w.dtype != torch.float32 or not w.is_contiguous():
                    raise RuntimeError("OPNet parameters must be contiguous fp32 on the input's device")
            return _OPNetTrainFunction.apply(self, boxes, *ws)
        B, T = int(boxes.shape[0]), int(boxes.shape[1])
        dev = boxes.device
        with torch.cuda.device(dev):
            packed = self._packed_weights(dev)
            stream = _stream_ptr(dev)
            # one workspace + graph per (shape, device, stream): forwards enqueued on different HIP
            # streams run concurrently (the step kernel leaves most of a CU idle at small batches)
            key = (B, T, dev.index if dev.index is not None else torch.cuda.current_device(), stream)
            if key not in self._plans:
                nbytes = lib.opnet_workspace_bytes(B, T, self._h1, self._h2)
                if nbytes == 0:
                    _lib.check(-2, "opnet_workspace_bytes")
                ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
                plan = _lib.c_void_p()
                _lib.check(lib.opnet_plan_create(_lib.ctypes.byref(plan), B, T, self._h1, self._h2),
                           "opnet_plan_create")
                self._plans[key] = (plan, ws)
            plan, ws = self._plans[key]
            y = torch.empty((B, T, 4), dtype=torch.float32, device=dev)
            logits = torch.empty((B, 15, T), dtype=torch.float32, device=dev)
            if self.use_graph:
                rc = lib.opnet_plan_forward(plan, boxes.data_ptr(), packed.data_ptr(), y.data_ptr(),
                                            logits.data_ptr(), ws.data_ptr(), ws.numel(), stream)
                _lib.check(rc, "opnet_plan_forward")
            else:
                rc = lib.opnet_forward_f32(boxes.data_ptr(), packed.data_ptr(), y.data_ptr(),
                                           logits.data_ptr(), ws.data_ptr(), ws.numel(), B, T,
                                           self._h1, self._h2, stream)
                _lib.check(rc, "opnet_forward_f32")
        return y, logits

    def __del__(self):
        try:
            lib = _lib.load()
            for plan, _ in self._plans.values():
                lib.opnet_plan_destroy(plan)
        except Exception:
            pass
