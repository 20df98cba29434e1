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
from .launch_monitor import LaunchMonitor


class AbstractCaterModel(nn.Module):
    """reference learned_models.py:8-15"""

    def __init__(self, config: Dict[str, int]):
        super().__init__()
        self.config: Dict[str, int] = config
        self.max_objects_in_frame = 15
        self.bb_in_dim = 5
        self.bb_out_dim = 4

    # the persistent-launch bookkeeping of the models whose LSTM stack runs through a _LstmStackRunner (OPNet has its own)
    def launch_guard(self):
        r = getattr(self, "_runner", None)
        return r.launch_guard() if r is not None else None

    def training_step_aborted(self) -> bool:
        r = getattr(self, "_runner", None)
        return r.training_step_aborted() if r is not None else False


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
        # with a gradient bucket on the module (data-parallel training, parallel.GradBucket) the six weight gradients are
        # written straight into its flat buffer: the all-reduce then needs no gather copy
        # - but only while no parameter holds a gradient: a p.grad that is still set (zero_grad(set_to_none=False), gradient
        # accumulation, a second backward) aliases the same slices after GradBucket.collect(), and AccumulateGrad would then
        # compute p.grad += new on aliased memory (2 x new instead of old + new)
        bucket = getattr(module, "_grad_bucket", None)
        if bucket is not None and len(bucket.params) == len(ctx.wshapes) and bucket.flat.device == dev and \
                all(tuple(p.shape) == s for p, s in zip(bucket.params, ctx.wshapes)) and \
                all(p.grad is None for p in bucket.params):
            grads = [bucket.view(i) for i in range(len(ctx.wshapes))]
        else:
            grads = [torch.empty(s, dtype=torch.float32, device=dev) for s in ctx.wshapes]
        with torch.cuda.device(dev):
            rc = lib.opnet_train_backward_f32(grad_y.data_ptr(), module._tpacked.data_ptr(), module._tws.data_ptr(),
                                              module._tws.numel(), *(g.data_ptr() for g in grads), B, T,
                                              module._h1, module._h2, _stream_ptr(dev))
        _lib.check(rc, "opnet_train_backward_f32")
        # the abort words of the step's two persistent launches (sticky from the forward): mirrored to the host behind the
        # backward; training.finish_step / OPNet.training_step_aborted() look at them at the caller's next sync point
        off = lib.opnet_train_status_offset(B, T, module._h1, module._h2)
        if off != _lib.NO_OFFSET and lib.opnet_xcd4_enabled():
            module._monitor.watch(module._tws, off, module._note_training_abort, "opnet_xcd4_forward/backward (training step)")
        return (None, None) + tuple(grads)


class _OPNetMlpTrainFunction(torch.autograd.Function):
    """OPNetLstmMlp training: the OPNet history/BPTT machinery with relu(hidden_layer) in the video role."""

    @staticmethod
    def forward(ctx, module, boxes, *weights):
        lib = _lib.load()
        B, T = int(boxes.shape[0]), int(boxes.shape[1])
        dev = boxes.device
        h1, h2 = module._h1, module._h2
        with torch.cuda.device(dev):
            stream = _stream_ptr(dev)
            nbytes = lib.opnet_train_packed_weights_bytes(h1, h2)
            if nbytes == 0:
                _lib.check(-2, "opnet_train_packed_weights_bytes")
            if module._tpacked is None or module._tpacked.device != dev:
                module._tpacked = torch.zeros(nbytes // 4, dtype=torch.float32, device=dev)
                module._tscratch = torch.empty(4 * h2 * 6, dtype=torch.float32, device=dev)
            rc = lib.opnet_mlp_train_pack_weights_f32(*(w.data_ptr() for w in weights), module._tpacked.data_ptr(),
                                                      nbytes, module._tscratch.data_ptr(), h1, h2, stream)
            _lib.check(rc, "opnet_mlp_train_pack_weights_f32")
            key = (B, T, str(dev))
            if module._tws_key != key:
                wsb = lib.opnet_train_workspace_bytes(B, T, h1, h2)
                if wsb == 0:
                    _lib.check(-2, "opnet_train_workspace_bytes")
                module._tws = None
                module._tws = torch.empty(wsb, dtype=torch.uint8, device=dev)
                module._tws_key = key
            y = torch.empty((B, T, 4), dtype=torch.float32, device=dev)
            logits = torch.empty((B, 15, T), dtype=torch.float32, device=dev)
            rc = lib.opnet_mlp_train_forward_f32(boxes.data_ptr(), module._tpacked.data_ptr(), y.data_ptr(),
                                                 logits.data_ptr(), module._tws.data_ptr(), module._tws.numel(),
                                                 B, T, h1, h2, stream)
            _lib.check(rc, "opnet_mlp_train_forward_f32")
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
            raise RuntimeError("OPNetLstmMlp: backward() after another training forward - the saved history of "
                               "this forward has been overwritten (one history per module)")
        n_in = 2 + len(ctx.wshapes)
        if grad_y is None:
            return (None,) * n_in
        lib = _lib.load()
        B, T = ctx.shape
        dev = grad_y.device
        grad_y = grad_y.contiguous().float()
        g_ih1, g_hh1, g_sel, _, g_out = [torch.empty(s, dtype=torch.float32, device=dev) for s in ctx.wshapes]
        g_hid4 = torch.empty((4 * module._h2, 6), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            rc = lib.opnet_mlp_train_backward_f32(grad_y.data_ptr(), module._tpacked.data_ptr(),
                                                  module._tws.data_ptr(), module._tws.numel(), g_ih1.data_ptr(),
                                                  g_hh1.data_ptr(), g_sel.data_ptr(), g_hid4.data_ptr(),
                                                  g_out.data_ptr(), B, T, module._h1, module._h2, _stream_ptr(dev))
        _lib.check(rc, "opnet_mlp_train_backward_f32")
        return (None, None, g_ih1, g_hh1, g_sel, g_hid4[:module._h2].contiguous(), g_out)


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
        self._packed: Dict[int, Tuple[tuple, torch.Tensor]] = {}      # per stream: (weights key, packed image)
        self._plans: Dict[Tuple[int, int, int, int], Tuple[int, torch.Tensor]] = {}
        self._retired = []
        self._tpacked = None     # training: inference tiles + transposed tiles
        self._tws = None         # training workspace (one forward's history)
        self._tws_key = None
        self._train_gen = 0
        self.use_graph = os.environ.get("OPNET_HIP_EAGER", "0") != "1"
        # per-XCD persistent forward (include/opnet_hip.h): "auto" = batches of at least XCD_MIN_BATCH clips at the
        # reference sizes; "1" / "0" force it on / off
        self.use_xcd = os.environ.get("OPNET_XCD", "auto")
        self._xws: Dict[Tuple[int, int, int, int], torch.Tensor] = {}
        self._xcd_ok = None
        # one small request (up to XCD4_MAX_BATCH clips) as one persistent launch of 4-clip groups: "auto" / "1" / "0"
        self.use_xcd4 = os.environ.get("OPNET_XCD4", "auto")
        self._x4packed: Dict[int, Tuple[tuple, torch.Tensor]] = {}     # per stream: (weights key, packed image)
        self._x4ws: Dict[Tuple[int, int, int, int], torch.Tensor] = {}
        self._monitor = LaunchMonitor()      # abort words of the persistent launches (launch_monitor.py)
        self._train_aborted = False

    # -- aborted persistent launches -------------------------------------------------------------
    def verify_launches(self) -> int:
        """Wait for the persistent launches issued so far and re-run every aborted one through the launch chain into the
        output tensors it returned (healed in place); returns the number of aborted launches.  The drivers call this at
        the sync point they already have, before anything reads the outputs on the host."""
        return self._monitor.verify()

    def _note_training_abort(self) -> None:
        self._train_aborted = True

    def training_step_aborted(self) -> bool:
        """after a sync: did a persistent launch of the training steps issued so far give up?  (Their gradients are NaN and
        a guarded FusedAdam step left the weights untouched.)  Switches this process to the launch chain and clears the flag."""
        self._monitor.verify()
        bad, self._train_aborted = self._train_aborted, False
        if bad:
            _lib.load().opnet_xcd4_enable(0)
        return bad

    def launch_guard(self):
        """the abort word (a 1-element int32 view of the current training workspace) that the persistent launches of a
        training step raise; None: no such launches for this shape - FusedAdam's guard"""
        if self._tws is None or self._tws_key is None:
            return None
        B, T, _ = self._tws_key
        off = _lib.load().opnet_train_status_offset(B, T, self._h1, self._h2)
        return None if off == _lib.NO_OFFSET else self._tws[off:off + 4].view(torch.int32)

    def _redo_on_chain(self, boxes: torch.Tensor, y: torch.Tensor, logits: torch.Tensor) -> None:
        with torch.no_grad(), torch.cuda.device(boxes.device):
            y2, lg2 = self._forward_chain(boxes)
            y.copy_(y2)
            logits.copy_(lg2)

    # -- weights ------------------------------------------------------------------------------
    def _weights(self):
        return (self.object_to_track_LSTM.weight_ih_l0, self.object_to_track_LSTM.weight_hh_l0,
                self.object_to_track_prediction.weight, self.video_LSTM.weight_ih_l0,
                self.video_LSTM.weight_hh_l0, self.prediction_layer.weight)

    def _packed_weights(self, device: torch.device) -> torch.Tensor:
        """the packed image for launches on the CURRENT stream.  One image per stream (at most 8): forwards of this module
        may be in flight on several streams, and a re-pack after a weight update must not rewrite an image that another
        stream's earlier launches are still reading - a pack and the launches that read it are always ordered by their
        own stream (the 4-clip form keeps its images the same way)."""
        lib = _lib.load()
        ws = self._weights()
        stream = _stream_ptr(device)
        key = tuple((w.data_ptr(), w._version) for w in ws) + (str(device),)
        entry = self._packed.get(stream)
        if entry is None or entry[0] != key:
            for w in ws:
                if w.device != device or w.dtype != torch.float32 or not w.is_contiguous():
                    raise RuntimeError("OPNet parameters must be contiguous fp32 on the input's device "
                                       "(call model.to(device) first)")
            nbytes = lib.opnet_packed_weights_bytes(self._h1, self._h2)
            if nbytes == 0:
                _lib.check(-2, "opnet_packed_weights_bytes")
            if entry is None or entry[1].device != device:
                if len(self._packed) >= 8:
                    # (the evicted image may still be read by launches on ITS stream: the caching allocator hands a block
                    # back to the stream it was allocated on, so whatever reuses it is ordered behind them)
                    self._packed.pop(next(iter(self._packed)))
                buf = torch.empty(nbytes // 4, dtype=torch.float32, device=device)
            else:
                buf = entry[1]
            rc = lib.opnet_pack_weights_f32(*(w.data_ptr() for w in ws), buf.data_ptr(), nbytes, self._h1, self._h2, stream)
            _lib.check(rc, "opnet_pack_weights_f32")
            self._packed[stream] = (key, buf)
        return self._packed[stream][1]

    XCD_MIN_BATCH = 64       # measured: 38.3 k clips/s against 37.4 k through the launch chain at 64 clips, 74 k against 49 k at 128
    MAX_PLANS = 16           # (shape, device, stream) launch plans kept alive

    def _wants_xcd(self, B: int) -> bool:
        if self.use_xcd in ("0", 0, False) or (self._h1, self._h2) != (256, 512):
            return False
        if self._xcd_ok is None:       # all 8 XCDs x 32 CUs visible on this device? (not in a compute partition)
            self._xcd_ok = bool(_lib.load().opnet_xcd_supported(self._h1, self._h2))
        if not self._xcd_ok:
            return False
        return self.use_xcd in ("1", 1, True) or B >= self.XCD_MIN_BATCH

    XCD4_MAX_BATCH = 64      # 4-clip groups, one (two) per XCD: 0.76 ms up to 32 clips, 1.35 ms up to 64, against 1.2 / 1.6-1.7 ms
                             # through the launch chain and 1.65 ms for 64 clips on the 16-clip persistent forward

    def _wants_xcd4(self, B: int) -> bool:
        if self.use_xcd4 in ("0", 0, False) or self.use_xcd in ("0", "1", 0, 1, False, True) or (self._h1, self._h2) != (256, 512):
            return False                # (use_xcd "1" forces the 16-clip form, "0" the launch chain)
        if B > self.XCD4_MAX_BATCH:
            return False
        if self._xcd_ok is None:
            self._xcd_ok = bool(_lib.load().opnet_xcd_supported(self._h1, self._h2))
        return self._xcd_ok

    def _forward_xcd4(self, boxes: torch.Tensor, stream: int):
        """up to 32 clips as ONE persistent launch of 4-clip groups, one per XCD (csrc/opnet_xcd4_kernels.hip)"""
        lib = _lib.load()
        B, T, dev = int(boxes.shape[0]), int(boxes.shape[1]), boxes.device
        ws_ = self._weights()
        key = tuple((w.data_ptr(), w._version) for w in ws_) + (str(dev),)
        # one packed image per stream (a re-pack on one stream must not rewrite what another stream's launch is reading)
        entry = self._x4packed.get(stream) if isinstance(self._x4packed, dict) else None
        if entry is None or entry[0] != key:
            for w in ws_:
                if w.device != dev or w.dtype != torch.float32 or not w.is_contiguous():
                    raise RuntimeError("OPNet parameters must be contiguous fp32 on the input's device "
                                       "(call model.to(device) first)")
            nbytes = lib.opnet_xcd4_packed_weights_bytes(self._h1, self._h2)
            if nbytes == 0:
                _lib.check(-2, "opnet_xcd4_packed_weights_bytes")
            if not isinstance(self._x4packed, dict):
                self._x4packed = {}
            if entry is None or entry[1].device != dev:
                if len(self._x4packed) >= 4:
                    self._x4packed.pop(next(iter(self._x4packed)))
                buf = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
            else:
                buf = entry[1]
            _lib.check(lib.opnet_xcd4_pack_weights_f32(*(w.data_ptr() for w in ws_), buf.data_ptr(), nbytes,
                                                       self._h1, self._h2, stream), "opnet_xcd4_pack_weights_f32")
            self._x4packed[stream] = (key, buf)
        x4packed = self._x4packed[stream][1]
        wkey = (B, T, dev.index if dev.index is not None else torch.cuda.current_device(), stream)
        if wkey not in self._x4ws:
            nbytes = lib.opnet_xcd4_workspace_bytes(B, T, self._h1, self._h2)
            if nbytes == 0:
                _lib.check(-2, "opnet_xcd4_workspace_bytes")
            if len(self._x4ws) >= 8:
                self._x4ws.pop(next(iter(self._x4ws)))
            self._x4ws[wkey] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        ws = self._x4ws[wkey]
        y = torch.empty((B, T, 4), dtype=torch.float32, device=dev)
        logits = torch.empty((B, 15, T), dtype=torch.float32, device=dev)
        _lib.check(lib.opnet_xcd4_forward_f32(boxes.data_ptr(), x4packed.data_ptr(), y.data_ptr(), logits.data_ptr(),
                                              ws.data_ptr(), ws.numel(), B, T, self._h1, self._h2, stream),
                   "opnet_xcd4_forward_f32")
        self._monitor.watch(ws, lib.opnet_xcd4_status_offset(B, T, self._h1, self._h2),
                            lambda: self._redo_on_chain(boxes, y, logits), "opnet_xcd4_forward")
        return y, logits

    def _forward_xcd(self, boxes: torch.Tensor, packed: torch.Tensor, stream: int):
        """one persistent launch per chunk of opnet_xcd_max_batch() clips (csrc/opnet_xcd_kernels.hip)"""
        lib = _lib.load()
        B, T, dev = int(boxes.shape[0]), int(boxes.shape[1]), boxes.device
        y = torch.empty((B, T, 4), dtype=torch.float32, device=dev)
        logits = torch.empty((B, 15, T), dtype=torch.float32, device=dev)
        step = int(lib.opnet_xcd_max_batch())
        while step > 64 and lib.opnet_xcd_workspace_bytes(min(step, B), T, self._h1, self._h2) == 0:
            step //= 2                  # very long clips: the history of a full launch would exceed one buffer descriptor
        for lo in range(0, B, step):
            n = min(step, B - lo)
            # one workspace per (shape, device, stream), like the launch plans: forwards enqueued on different streams
            # must not share a history buffer
            key = (n, T, dev.index if dev.index is not None else torch.cuda.current_device(), stream)
            if key not in self._xws:
                nbytes = lib.opnet_xcd_workspace_bytes(n, T, self._h1, self._h2)
                if nbytes == 0:
                    _lib.check(-2, "opnet_xcd_workspace_bytes")
                if len(self._xws) >= 8:
                    # dropped at once, unlike the chain's plans (a hipGraph must be destroyed by hand, a tensor must not):
                    # the key holds the stream, the buffer was allocated on it, and the caching allocator only hands a block
                    # back to its own stream - whatever reuses it runs behind the launch that may still be using it
                    self._xws.pop(next(iter(self._xws)))
                self._xws[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            ws = self._xws[key]
            rc = lib.opnet_xcd_forward_f32(boxes[lo:lo + n].data_ptr(), packed.data_ptr(), y[lo:lo + n].data_ptr(),
                                           logits[lo:lo + n].data_ptr(), ws.data_ptr(), ws.numel(), n, T,
                                           self._h1, self._h2, stream)
            _lib.check(rc, "opnet_xcd_forward_f32")
            self._monitor.watch(ws, 0, lambda b=boxes[lo:lo + n], yy=y[lo:lo + n], ll=logits[lo:lo + n]:
                                self._redo_on_chain(b, yy, ll), "opnet_xcd_forward")
        return y, logits

    def forward_requests(self, requests):
        """The forward of several request tensors [b_r, T, 15, 6] as ONE batch (serving.ReasonerServer): equal to
        `self(torch.cat(requests))`, without the concatenation copy when the batch runs as one 16-clip persistent launch
        (opnet_xcd_forward_multi_f32 reads the requests where they lie)."""
        lib = _lib.load()
        B = sum(int(r.shape[0]) for r in requests)
        ok = (len(requests) > 1 and len(requests) <= 64 and not torch.is_grad_enabled() and all(r.is_cuda for r in requests)
              and not self._wants_xcd4(B) and self._wants_xcd(B) and B <= int(lib.opnet_xcd_max_batch())
              and all(r.dim() == 4 and r.shape[1:] == requests[0].shape[1:] and r.device == requests[0].device for r in requests))
        if not ok:
            return self(requests[0] if len(requests) == 1 else torch.cat(list(requests), dim=0))
        reqs = [r if (r.is_contiguous() and r.dtype == torch.float32) else r.contiguous().float() for r in requests]
        T, dev = int(reqs[0].shape[1]), reqs[0].device
        if reqs[0].shape[2] != 15 or reqs[0].shape[3] != 6:
            raise ValueError(f"boxes must be [B, T, 15, 6], got {tuple(reqs[0].shape)}")
        with torch.cuda.device(dev):
            packed = self._packed_weights(dev)
            stream = _stream_ptr(dev)
            nbytes = lib.opnet_xcd_workspace_bytes(B, T, self._h1, self._h2)
            if nbytes == 0:     # a history beyond one buffer descriptor: the chunked path
                return self(torch.cat(reqs, dim=0))
            key = (B, T, dev.index if dev.index is not None else torch.cuda.current_device(), stream)
            if key not in self._xws:
                if len(self._xws) >= 8:
                    self._xws.pop(next(iter(self._xws)))
                self._xws[key] = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            ws = self._xws[key]
            y = torch.empty((B, T, 4), dtype=torch.float32, device=dev)
            logits = torch.empty((B, 15, T), dtype=torch.float32, device=dev)
            n = len(reqs)
            ptrs = (_lib.c_void_p * n)(*[r.data_ptr() for r in reqs])
            counts = (_lib.ctypes.c_int * n)(*[int(r.shape[0]) for r in reqs])
            _lib.check(lib.opnet_xcd_forward_multi_f32(ptrs, counts, n, packed.data_ptr(), y.data_ptr(), logits.data_ptr(),
                                                       ws.data_ptr(), ws.numel(), T, self._h1, self._h2, stream),
                       "opnet_xcd_forward_multi_f32")
            self._monitor.watch(ws, 0, lambda: self._redo_on_chain(torch.cat(reqs, dim=0), y, logits), "opnet_xcd_forward")
        return y, logits

    def xcd_status(self):
        """{abort code, block, phase} of the last persistent launches (synchronises); code 0 = completed"""
        return {k: ws[:12].view(torch.int32).tolist() for k, ws in self._xws.items()}

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
            if self._wants_xcd4(B):
                return self._forward_xcd4(boxes, _stream_ptr(dev))
            packed = self._packed_weights(dev)
            stream = _stream_ptr(dev)
            if self._wants_xcd(B):
                return self._forward_xcd(boxes, packed, stream)
            return self._forward_chain(boxes)

    def _forward_chain(self, boxes: torch.Tensor):
        """the launch-per-step form (csrc/opnet_kernels.hip): one hipGraph of T + 3 step launches"""
        lib = _lib.load()
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
                # least-recently-used bound: a server that sees many (shape, stream) pairs must not keep a hipGraph and
                # a workspace for each of them forever.  The evicted plan may still have work in flight on its stream:
                # park it until the next eviction instead of destroying it under the GPU.
                while len(self._plans) >= self.MAX_PLANS:
                    old_key = next(iter(self._plans))
                    self._retired.append(self._plans.pop(old_key))
                    if len(self._retired) > self.MAX_PLANS:
                        torch.cuda.synchronize(dev)
                        for old_plan, _ in self._retired:
                            lib.opnet_plan_destroy(old_plan)
                        self._retired.clear()
                self._plans[key] = (plan, ws)
            else:
                self._plans[key] = self._plans.pop(key)      # move to the most-recently-used end
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
            for plan, _ in list(self._plans.values()) + list(self._retired):
                lib.opnet_plan_destroy(plan)
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------
# sibling reasoners (reference learned_models.py:55-197) - inference through the HIP library
# ------------------------------------------------------------------------------------------------
def _check_input(module: nn.Module, x: torch.Tensor, feat: int):
    if not x.is_cuda:
        raise RuntimeError(f"objectpermanence_amd.{type(module).__name__} runs on MI355X only: move the input "
                           "(and the model) to a ROCm device; there is no CPU fallback")
    if x.dim() != 4 or x.shape[2] != 15 or x.shape[3] != feat:
        raise ValueError(f"input must be [B, T, 15, {feat}], got {tuple(x.shape)}")


def _wants_grad(module: nn.Module) -> bool:
    return torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters())


class _StackTrainFunction(torch.autograd.Function):
    """autograd bridge of the stacked LSTM: opseq_lstm_stack_train_forward_f32 / _backward_f32"""

    @staticmethod
    def forward(ctx, runner, x, *weights):
        lib = _lib.load()
        L, KX, H = runner.L, runner.KX, runner.H
        B, T, dev = int(x.shape[0]), int(x.shape[1]), x.device
        stream = _stream_ptr(dev)
        nbytes = lib.opseq_lstm_stack_train_packed_bytes(L, KX, H)
        if nbytes == 0:
            _lib.check(-2, "opseq_lstm_stack_train_packed_bytes")
        if runner.tpacked is None or runner.tpacked.device != dev:
            runner.tpacked = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
        arr = _lib.c_void_p * L
        ih = arr(*[w.data_ptr() for w in weights[:L]])
        hh = arr(*[w.data_ptr() for w in weights[L:2 * L]])
        _lib.check(lib.opseq_lstm_stack_train_pack_weights_f32(ih, hh, weights[2 * L].data_ptr(), runner.tpacked.data_ptr(),
                                                               nbytes, L, KX, H, stream), "opseq_lstm_stack_train_pack_weights_f32")
        key = (B, T, str(dev))
        if runner.tws_key != key:
            wsb = lib.opseq_lstm_stack_train_workspace_bytes(B, T, L, KX, H)
            if wsb == 0:
                _lib.check(-2, "opseq_lstm_stack_train_workspace_bytes")
            runner.tws = None
            runner.tws = torch.empty(wsb, dtype=torch.uint8, device=dev)
            runner.tws_key = key
        y = torch.empty((B, T, 4), dtype=torch.float32, device=dev)
        _lib.check(lib.opseq_lstm_stack_train_forward_f32(x.data_ptr(), runner.tpacked.data_ptr(), y.data_ptr(),
                                                          runner.tws.data_ptr(), runner.tws.numel(), B, T, L, KX, H, stream),
                   "opseq_lstm_stack_train_forward_f32")
        runner.train_gen += 1
        ctx.runner, ctx.gen, ctx.shape = runner, runner.train_gen, (B, T)
        ctx.wshapes = [tuple(w.shape) for w in weights]
        ctx.need_dx = x.requires_grad
        # the persistent training forward (H = 512 stacks): its abort words are mirrored to the host behind the launch
        off = lib.opseq_lstm_stack_train_status_offset(B, T, L, KX, H)
        if off != _lib.NO_OFFSET and lib.opseq_xcd_supported(L, KX, H):
            runner._monitor.watch(runner.tws, off, runner._note_training_abort, "seqx_forward (training)")
        return y

    @staticmethod
    def backward(ctx, grad_y):
        runner = ctx.runner
        if ctx.gen != runner.train_gen:
            raise RuntimeError("backward() after another training forward - the saved history has been overwritten")
        lib = _lib.load()
        L, KX, H = runner.L, runner.KX, runner.H
        B, T = ctx.shape
        dev = grad_y.device
        grad_y = grad_y.contiguous().float()
        grads = [torch.empty(s, dtype=torch.float32, device=dev) for s in ctx.wshapes]
        dx = torch.empty((B, T, KX), dtype=torch.float32, device=dev) if ctx.need_dx else None
        arr = _lib.c_void_p * L
        gih = arr(*[g.data_ptr() for g in grads[:L]])
        ghh = arr(*[g.data_ptr() for g in grads[L:2 * L]])
        with torch.cuda.device(dev):
            rc = lib.opseq_lstm_stack_train_backward_f32(grad_y.data_ptr(), runner.tpacked.data_ptr(), runner.tws.data_ptr(),
                                                         runner.tws.numel(), gih, ghh, grads[2 * L].data_ptr(),
                                                         None if dx is None else dx.data_ptr(), B, T, L, KX, H,
                                                         _stream_ptr(dev))
        _lib.check(rc, "opseq_lstm_stack_train_backward_f32")
        # the persistent reverse recurrence (csrc/seq_xcd_bwd_kernels.hip) raises the same status words as the forward when it gives
        # up (its gradients are then NaN and the guarded optimiser leaves the weights alone)
        off = lib.opseq_lstm_stack_train_status_offset(B, T, L, KX, H)
        if off != _lib.NO_OFFSET and lib.opseq_xcd_supported(L, KX, H):
            with torch.cuda.device(dev):
                runner._monitor.watch(runner.tws, off, runner._note_training_abort, "seqx_backward (training)")
        return (None, dx) + tuple(grads)


class _SlotEmbedFunction(torch.autograd.Function):
    """relu(boxes_linear(x)) per slot with its weight gradient (the input boxes never need a gradient)"""

    @staticmethod
    def forward(ctx, x, weight, nslots_out):
        lib = _lib.load()
        ntok, F = int(x.shape[0] * x.shape[1]), int(weight.shape[0])
        out = torch.empty((x.shape[0], x.shape[1], nslots_out * F), dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            _lib.check(lib.opseq_slot_embed_relu_f32(x.data_ptr(), weight.data_ptr(), out.data_ptr(), ntok, nslots_out, F,
                                                     _stream_ptr(x.device)), "opseq_slot_embed_relu_f32")
        ctx.save_for_backward(x, out)
        ctx.dims = (ntok, nslots_out, F)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        x, out = ctx.saved_tensors
        ntok, nslots_out, F = ctx.dims
        dW = torch.empty((F, 5), dtype=torch.float32, device=x.device)
        dout = dout.contiguous().float()
        with torch.cuda.device(x.device):
            nws = int(lib.opseq_slot_embed_bwd_workspace_bytes(ntok, nslots_out, F))
            ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=x.device)      # per-workgroup partial sums (<= 5 MB)
            _lib.check(lib.opseq_slot_embed_relu_bwd_ws_f32(x.data_ptr(), out.data_ptr(), dout.data_ptr(), dW.data_ptr(), ntok,
                                                            nslots_out, F, ws.data_ptr(), ws.numel(), _stream_ptr(x.device)),
                       "opseq_slot_embed_relu_bwd_ws_f32")
        return None, dW, None


def _weights_key(ws, dev):
    return tuple((w.data_ptr(), w._version) for w in ws) + (str(dev),)


class _LstmStackRunner:
    """Packs L stacked LSTM layers + head once per weight version and runs opseq_lstm_stack_forward_f32."""

    def __init__(self, layers: int, kx: int, hidden: int):
        self.L, self.KX, self.H = layers, kx, hidden
        self.packed = None
        self.key = None
        self.ws = {}
        self.tpacked, self.tws, self.tws_key, self.train_gen = None, None, None, 0
        # the whole stack as ONE persistent launch (csrc/seq_xcd_kernels.hip): "auto" = whenever the library supports the
        # shape on this device (H = 512; the reference's three stacked reasoners) and the batch fits one launch; "0" = never
        self.use_xcd = os.environ.get("OPSEQ_XCD", "auto")
        self._xpacked: Dict[int, Tuple[tuple, torch.Tensor]] = {}     # per stream: (weights key, register image)
        self._cpacked: Dict[int, Tuple[tuple, torch.Tensor]] = {}     # per stream: (weights key, the launch chain's packed image)
        self._xws: Dict[tuple, torch.Tensor] = {}
        self._monitor = LaunchMonitor()
        self.xcd_launches = 0            # statistics: forwards that ran as one persistent launch
        # ... and its throughput form (csrc/seq_xcdt_kernels.hip): "auto" = batches of at least XCDT_MIN_BATCH clips, "1" = every
        # supported batch, "0" = never
        self.use_xcdt = os.environ.get("OPSEQ_XCDT", "auto")
        # the first batch the 4-clip form needs one more round for (it carries 32 clips a round with one layer, 16 with two): measured
        # (tools/xcdt_threshold_probe.py) 65 clips 1.43 -> 1.20 ms, two layers 33 clips 3.21 -> 3.03 ms; at 64 / 32 the 4-clip form wins
        self.XCDT_MIN_BATCH = int(os.environ.get("OPSEQ_XCDT_MIN_BATCH", 65 if layers == 1 else 33))
        self._tpacked: Dict[int, Tuple[tuple, torch.Tensor]] = {}
        self._tws: Dict[tuple, torch.Tensor] = {}
        self.xcdt_launches = 0

    def _note_training_abort(self) -> None:
        self._train_aborted = True

    def training_step_aborted(self) -> bool:
        """after a sync: did the persistent forward of a training step give up?  (y and the loss are NaN, the guarded
        FusedAdam left the weights alone.)  Switches this process to the launch chain and clears the flag."""
        self._monitor.verify()
        bad, self._train_aborted = getattr(self, "_train_aborted", False), False
        if bad:
            _lib.load().opseq_xcd_enable(0)
        return bad

    def launch_guard(self):
        if self.tws is None or self.tws_key is None:
            return None
        B, T, _ = self.tws_key
        off = _lib.load().opseq_lstm_stack_train_status_offset(B, T, self.L, self.KX, self.H)
        return None if off == _lib.NO_OFFSET else self.tws[off:off + 4].view(torch.int32)

    def _wants_xcd(self, B: int, T: Optional[int] = None) -> bool:
        """the persistent launch takes (B, T) when the shape is supported on this device, the batch fits one launch and -
        T given - its workspace stays below 2 GiB (opseq_xcd_workspace_bytes says 0 beyond: 32-bit offsets in the kernel;
        e.g. NonLinearLstm's hoisted input products at long T); everything else runs the launch chain"""
        if self.use_xcd in ("0", 0, False):
            return False
        lib = _lib.load()
        if not (bool(lib.opseq_xcd_supported(self.L, self.KX, self.H)) and B <= int(lib.opseq_xcd_max_batch(self.L))):
            return False
        return T is None or int(lib.opseq_xcd_workspace_bytes(B, T, self.L, self.KX, self.H)) > 0

    def _run_xcd(self, x: torch.Tensor, ws_list, head: "LinearWeight") -> torch.Tensor:
        lib = _lib.load()
        dev = x.device
        B, T = int(x.shape[0]), int(x.shape[1])
        stream = _stream_ptr(dev)
        key = _weights_key(ws_list, dev)
        entry = self._xpacked.get(stream)
        if entry is None or entry[0] != key:
            for w in ws_list:
                if w.device != dev or w.dtype != torch.float32 or not w.is_contiguous():
                    raise RuntimeError("parameters must be contiguous fp32 on the input's device")
            nbytes = lib.opseq_xcd_packed_bytes(self.L, self.KX, self.H)
            if nbytes == 0:
                _lib.check(-2, "opseq_xcd_packed_bytes")
            if entry is None or entry[1].device != dev:
                if len(self._xpacked) >= 4:
                    self._xpacked.pop(next(iter(self._xpacked)))
                buf = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
            else:
                buf = entry[1]
            arr = _lib.c_void_p * self.L
            ih = arr(*[w.data_ptr() for w in ws_list[:self.L]])
            hh = arr(*[w.data_ptr() for w in ws_list[self.L:2 * self.L]])
            _lib.check(lib.opseq_xcd_pack_weights_f32(ih, hh, buf.data_ptr(), nbytes, self.L, self.KX, self.H, stream),
                       "opseq_xcd_pack_weights_f32")
            self._xpacked[stream] = (key, buf)
        packed = self._xpacked[stream][1]
        wkey = (B, T, str(dev), stream)
        if wkey not in self._xws:
            nb = lib.opseq_xcd_workspace_bytes(B, T, self.L, self.KX, self.H)
            if nb == 0:
                _lib.check(-2, "opseq_xcd_workspace_bytes")
            if len(self._xws) >= 4:
                self._xws.pop(next(iter(self._xws)))      # (stream-keyed: see OPNet._forward_xcd on dropping it at once)
            self._xws[wkey] = torch.empty(nb, dtype=torch.uint8, device=dev)
        ws = self._xws[wkey]
        y = torch.empty((B, T, 4), dtype=torch.float32, device=dev)
        _lib.check(lib.opseq_xcd_forward_f32(x.data_ptr(), packed.data_ptr(), head.weight.data_ptr(), y.data_ptr(), ws.data_ptr(),
                                             ws.numel(), B, T, self.L, self.KX, self.H, stream), "opseq_xcd_forward_f32")

        def redo():          # the launch gave up: the same batch through the launch-per-step chain, into the same y
            with torch.no_grad(), torch.cuda.device(dev):
                y.copy_(self._run_chain(x, ws_list, head))

        self._monitor.watch(ws, lib.opseq_xcd_status_offset(B, T, self.L, self.KX, self.H), redo, "seqx_forward")
        self.xcd_launches += 1
        return y

    def run_train(self, x: torch.Tensor, lstm: "LSTMWeights", head: "LinearWeight") -> torch.Tensor:
        ws_list = [getattr(lstm, f"weight_ih_l{l}") for l in range(self.L)] + \
                  [getattr(lstm, f"weight_hh_l{l}") for l in range(self.L)] + [head.weight]
        for w in ws_list:
            if w.device != x.device or w.dtype != torch.float32 or not w.is_contiguous():
                raise RuntimeError("parameters must be contiguous fp32 on the input's device")
        return _StackTrainFunction.apply(self, x.contiguous(), *ws_list)

    # Batches from XCDT_MIN_BATCH clips on run the throughput form (16-clip groups, csrc/seq_xcdt_kernels.hip); below, the 4-clip
    # latency form keeps lone / small requests (and its bit-identity between a served request and its lone forward).  Measured
    # cross-over (one forward, T = 300, MI355X): L = 1: 64 clips 1.28 ms against 1.08, 128 clips 1.30 against 1.95 (the latency form
    # carries 32 clips per 0.52-ms round, the throughput form up to 128 in 1.28 ms) -> 80; L = 2: the latency form carries 16 clips
    # per 0.67-ms round, the throughput form up to 64 in 1.67 ms -> 40
    XCDT_MIN_BATCH = 64          # (class default; per instance by layer count, see __init__)

    def _wants_xcdt(self, B: int, T: int) -> bool:
        if self.use_xcdt in ("0", 0, False) or self.use_xcd in ("0", 0, False):
            return False
        lib = _lib.load()
        if not bool(lib.opseq_xcdt_supported(self.L, self.KX, self.H)) or int(lib.opseq_xcdt_max_batch(T, self.L, self.KX, self.H)) < 16:
            return False
        return self.use_xcdt in ("1", 1, True) or B >= self.XCDT_MIN_BATCH

    def _run_xcdt(self, x: torch.Tensor, ws_list, head: "LinearWeight") -> torch.Tensor:
        """one persistent launch of 16-clip groups per chunk of opseq_xcdt_max_batch clips"""
        lib = _lib.load()
        dev = x.device
        B, T = int(x.shape[0]), int(x.shape[1])
        stream = _stream_ptr(dev)
        key = _weights_key(ws_list, dev)
        entry = self._tpacked.get(stream)
        if entry is None or entry[0] != key:
            for w in ws_list:
                if w.device != dev or w.dtype != torch.float32 or not w.is_contiguous():
                    raise RuntimeError("parameters must be contiguous fp32 on the input's device")
            nbytes = lib.opseq_xcdt_packed_bytes(self.L, self.KX, self.H)
            if nbytes == 0:
                _lib.check(-2, "opseq_xcdt_packed_bytes")
            if entry is None or entry[1].device != dev:
                if len(self._tpacked) >= 4:
                    self._tpacked.pop(next(iter(self._tpacked)))
                buf = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
            else:
                buf = entry[1]
            arr = _lib.c_void_p * self.L
            ih = arr(*[w.data_ptr() for w in ws_list[:self.L]])
            hh = arr(*[w.data_ptr() for w in ws_list[self.L:2 * self.L]])
            _lib.check(lib.opseq_xcdt_pack_weights_f32(ih, hh, buf.data_ptr(), nbytes, self.L, self.KX, self.H, stream),
                       "opseq_xcdt_pack_weights_f32")
            self._tpacked[stream] = (key, buf)
        packed = self._tpacked[stream][1]
        step = int(lib.opseq_xcdt_max_batch(T, self.L, self.KX, self.H))
        y = torch.empty((B, T, 4), dtype=torch.float32, device=dev)
        for b0 in range(0, B, step):
            n = min(step, B - b0)
            wkey = (n, T, str(dev), stream)
            if wkey not in self._tws:
                nb = lib.opseq_xcdt_workspace_bytes(n, T, self.L, self.KX, self.H)
                if nb == 0:
                    _lib.check(-2, "opseq_xcdt_workspace_bytes")
                if len(self._tws) >= 4:
                    self._tws.pop(next(iter(self._tws)))
                self._tws[wkey] = torch.empty(nb, dtype=torch.uint8, device=dev)
            ws = self._tws[wkey]
            xc, yc = x[b0:b0 + n], y[b0:b0 + n]
            _lib.check(lib.opseq_xcdt_forward_f32(xc.data_ptr(), packed.data_ptr(), head.weight.data_ptr(), yc.data_ptr(), ws.data_ptr(),
                                                  ws.numel(), n, T, self.L, self.KX, self.H, stream), "opseq_xcdt_forward_f32")

            def redo(xc=xc, yc=yc):      # the launch gave up: the same clips through the launch-per-step chain, into the same y
                with torch.no_grad(), torch.cuda.device(dev):
                    yc.copy_(self._run_chain(xc, ws_list, head))

            self._monitor.watch(ws, lib.opseq_xcdt_status_offset(n, T, self.L, self.KX, self.H), redo, "seqt_forward")
            self.xcdt_launches += 1
        return y

    def engine(self, B: int, T: int) -> str:
        """what a forward of B clips x T frames runs on: "t" the persistent launch of 16-clip groups (csrc/seq_xcdt_kernels.hip),
        "x" the persistent launch of 4-clip groups (csrc/seq_xcd_kernels.hip), "c" one launch per time step"""
        if self._wants_xcdt(B, T):
            return "t"
        return "x" if self._wants_xcd(B, T) else "c"

    def run(self, x: torch.Tensor, lstm: "LSTMWeights", head: "LinearWeight", engine: Optional[str] = None) -> torch.Tensor:
        """engine: None = by this batch's shape; a caller that merged independent requests and wants each of them computed as it
        would be alone passes the lone request's engine"""
        ws_list = [getattr(lstm, f"weight_ih_l{l}") for l in range(self.L)] + \
                  [getattr(lstm, f"weight_hh_l{l}") for l in range(self.L)] + [head.weight]
        if engine is None:
            engine = self.engine(int(x.shape[0]), int(x.shape[1]))
        if engine == "t":
            return self._run_xcdt(x, ws_list, head)
        if engine == "x":
            # a caller may ask for the lone request's 4-clip engine on a merged pass that is larger than one such launch carries
            # (OPSEQ_XCDT_MIN_BATCH raised above opseq_xcd_max_batch: ADVICE round 5): clips are independent in the stack, so the
            # pass runs as several launches of whole 4-clip groups - every clip as in its lone forward
            cap = int(_lib.load().opseq_xcd_max_batch(self.L))
            if int(x.shape[0]) > cap:
                return torch.cat([self._run_xcd(x[i:i + cap].contiguous(), ws_list, head) for i in range(0, int(x.shape[0]), cap)])
            return self._run_xcd(x, ws_list, head)
        return self._run_chain(x, ws_list, head)

    def _run_chain(self, x: torch.Tensor, ws_list, head: "LinearWeight") -> torch.Tensor:
        """one launch per time step (csrc/seq_kernels.hip lstm_stack_step), T + 2 L - 1 launches as one hipGraph"""
        lib = _lib.load()
        dev = x.device
        B, T = int(x.shape[0]), int(x.shape[1])
        stream = _stream_ptr(dev)
        key = _weights_key(ws_list, dev)
        # packed image and workspace PER STREAM (as _run_xcd): a server issues a segmented model's passes on two side streams in
        # turn - a pass on stream B must neither read an image stream A is still packing nor share A's workspace
        entry = self._cpacked.get(stream)
        if entry is None or entry[0] != key:
            for w in ws_list:
                if w.device != dev or w.dtype != torch.float32 or not w.is_contiguous():
                    raise RuntimeError("parameters must be contiguous fp32 on the input's device")
            nbytes = lib.opseq_lstm_stack_packed_bytes(self.L, self.KX, self.H)
            if nbytes == 0:
                _lib.check(-2, "opseq_lstm_stack_packed_bytes")
            if entry is None or entry[1].device != dev:
                if len(self._cpacked) >= 4:
                    self._cpacked.pop(next(iter(self._cpacked)))
                buf = torch.empty(nbytes // 4, dtype=torch.float32, device=dev)
            else:
                buf = entry[1]
            arr = _lib.c_void_p * self.L
            ih = arr(*[w.data_ptr() for w in ws_list[:self.L]])
            hh = arr(*[w.data_ptr() for w in ws_list[self.L:2 * self.L]])
            rc = lib.opseq_lstm_stack_pack_weights_f32(ih, hh, head.weight.data_ptr(), buf.data_ptr(), nbytes,
                                                       self.L, self.KX, self.H, stream)
            _lib.check(rc, "opseq_lstm_stack_pack_weights_f32")
            self._cpacked[stream] = (key, buf)
        self.packed, self.key = self._cpacked[stream][1], key
        wkey = (B, T, str(dev), stream)
        if wkey not in self.ws:
            nb = lib.opseq_lstm_stack_workspace_bytes(B, T, self.L, self.KX, self.H)
            if nb == 0:
                _lib.check(-2, "opseq_lstm_stack_workspace_bytes")
            # one shape per stream at a time (the cached hipGraph is keyed by the workspace: a new buffer is a new graph)
            self.ws = {k: v for k, v in self.ws.items() if k[3] != stream}
            if len(self.ws) >= 4:
                self.ws.pop(next(iter(self.ws)))
            self.ws[wkey] = torch.empty(nb, dtype=torch.uint8, device=dev)
        ws = self.ws[wkey]
        y = torch.empty((B, T, 4), dtype=torch.float32, device=dev)
        fwd = lib.opseq_lstm_stack_forward_f32 if os.environ.get("OPNET_HIP_EAGER", "0") == "1" \
            else lib.opseq_lstm_stack_forward_graph_f32
        rc = fwd(x.data_ptr(), self.packed.data_ptr(), y.data_ptr(), ws.data_ptr(), ws.numel(), B, T, self.L,
                 self.KX, self.H, stream)
        _lib.check(rc, "opseq_lstm_stack_forward_f32")
        return y


class BaselineLstm(AbstractCaterModel):
    """reference learned_models.py:92-118. forward(x [B,T,15,5]) -> y_boxes [B,T,4]."""

    def __init__(self, config: Dict[str, int]):
        super().__init__(config)
        h = config["videos_hidden_dim"]
        self.video_LSTM = LSTMWeights(self.max_objects_in_frame * self.bb_in_dim, h)
        self.predictions_layer = LinearWeight(h, self.bb_out_dim)
        self._runner = _LstmStackRunner(1, self.max_objects_in_frame * self.bb_in_dim, h)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _check_input(self, x, 5)
        x = x.contiguous().float()
        with torch.cuda.device(x.device):
            flat = x.view(x.shape[0], x.shape[1], -1)
            if _wants_grad(self):
                return self._runner.run_train(flat, self.video_LSTM, self.predictions_layer)
            return self._runner.run(flat, self.video_LSTM, self.predictions_layer)


class NonLinearLstm(AbstractCaterModel):
    """reference learned_models.py:121-151: relu(Linear 5->F) per slot -> 2-layer LSTM -> Linear."""

    def __init__(self, config: Dict[str, int]):
        super().__init__(config)
        f, h = config["boxes_features_dim"], config["videos_hidden_dim"]
        self.boxes_linear = LinearWeight(self.bb_in_dim, f)
        self.video_LSTM = LSTMWeights(self.max_objects_in_frame * f, h, num_layers=2)
        self.predictions_layer = LinearWeight(h, self.bb_out_dim)
        self._f = f
        self._runner = _LstmStackRunner(2, self.max_objects_in_frame * f, h)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _check_input(self, x, 5)
        lib = _lib.load()
        x = x.contiguous().float()
        B, T = int(x.shape[0]), int(x.shape[1])
        with torch.cuda.device(x.device):
            if _wants_grad(self):
                feats = _SlotEmbedFunction.apply(x, self.boxes_linear.weight, 15)
                return self._runner.run_train(feats, self.video_LSTM, self.predictions_layer)
            feats = torch.empty((B, T, 15 * self._f), dtype=torch.float32, device=x.device)
            rc = lib.opseq_slot_embed_relu_f32(x.data_ptr(), self.boxes_linear.weight.data_ptr(), feats.data_ptr(),
                                               B * T, 15, self._f, _stream_ptr(x.device))
            _lib.check(rc, "opseq_slot_embed_relu_f32")
            return self._runner.run(feats, self.video_LSTM, self.predictions_layer)


class _EncoderLayerTrainFunction(torch.autograd.Function):
    """autograd bridge of one nn.TransformerEncoderLayer in training: opseq_encoder_layer_train_forward_f32 /
    _backward_f32.  The layer's activations live in a per-call `saved` buffer, the scratch is the owner module's."""

    @staticmethod
    def forward(ctx, z, owner, p_drop, seed, *weights):
        lib = _lib.load()
        S, E = int(z.shape[0]), int(z.shape[1])
        nhead, ffn, dev = owner._nhead, owner.FFN, z.device
        nsaved = lib.opseq_encoder_train_saved_bytes(S, E, nhead, ffn)
        nscr = lib.opseq_encoder_train_scratch_bytes(S, E, nhead, ffn)
        if nsaved == 0 or nscr == 0:
            _lib.check(-2, "opseq_encoder_train_saved_bytes")
        if owner._tscratch is None or owner._tscratch.numel() < nscr or owner._tscratch.device != dev:
            owner._tscratch = None
            owner._tscratch = torch.empty(nscr, dtype=torch.uint8, device=dev)
        saved = torch.empty(nsaved, dtype=torch.uint8, device=dev)
        z = z.contiguous()
        z_out = torch.empty_like(z)
        with torch.cuda.device(dev):
            rc = lib.opseq_encoder_layer_train_forward_f32(
                z.data_ptr(), z_out.data_ptr(), *(w.data_ptr() for w in weights), saved.data_ptr(), nsaved,
                owner._tscratch.data_ptr(), owner._tscratch.numel(), S, E, nhead, ffn, float(p_drop), int(seed), _stream_ptr(dev))
        _lib.check(rc, "opseq_encoder_layer_train_forward_f32")
        ctx.save_for_backward(*weights)
        ctx.owner, ctx.saved, ctx.meta = owner, saved, (S, E, nhead, ffn, float(p_drop), int(seed))
        return z_out

    @staticmethod
    def backward(ctx, dz_out):
        lib = _lib.load()
        weights = ctx.saved_tensors
        S, E, nhead, ffn, p_drop, seed = ctx.meta
        owner, dev = ctx.owner, dz_out.device
        dz_out = dz_out.contiguous().float()
        dz_in = torch.empty_like(dz_out)
        grads = [torch.empty_like(w) for w in weights]
        in_w, _in_b, out_w, _out_b, l1_w, _l1_b, l2_w, _l2_b, n1_w, _n1_b, n2_w, _n2_b = weights
        with torch.cuda.device(dev):
            rc = lib.opseq_encoder_layer_train_backward_f32(
                dz_out.data_ptr(), dz_in.data_ptr(), in_w.data_ptr(), out_w.data_ptr(), l1_w.data_ptr(), l2_w.data_ptr(),
                n1_w.data_ptr(), n2_w.data_ptr(), *(g.data_ptr() for g in grads), ctx.saved.data_ptr(), ctx.saved.numel(),
                owner._tscratch.data_ptr(), owner._tscratch.numel(), S, E, nhead, ffn, p_drop, seed, _stream_ptr(dev))
        _lib.check(rc, "opseq_encoder_layer_train_backward_f32")
        ctx.saved = None
        return (dz_in, None, None, None) + tuple(grads)


class _EncoderLayerWeights(nn.Module):
    """Parameter holder with nn.TransformerEncoderLayer's parameter names (torch defaults for init)."""

    class _Attn(nn.Module):
        def __init__(self, e):
            super().__init__()
            self.in_proj_weight = nn.Parameter(torch.empty(3 * e, e))
            self.in_proj_bias = nn.Parameter(torch.zeros(3 * e))
            self.out_proj = nn.Linear(e, e)      # holder only: never called
            nn.init.xavier_uniform_(self.in_proj_weight)
            nn.init.zeros_(self.out_proj.bias)

    def __init__(self, e: int, ffn: int):
        super().__init__()
        self.self_attn = self._Attn(e)
        self.linear1 = nn.Linear(e, ffn)         # holders only: never called
        self.linear2 = nn.Linear(ffn, e)
        self.norm1 = nn.LayerNorm(e)
        self.norm2 = nn.LayerNorm(e)

    def tensors(self):
        return [self.self_attn.in_proj_weight, self.self_attn.in_proj_bias, self.self_attn.out_proj.weight,
                self.self_attn.out_proj.bias, self.linear1.weight, self.linear1.bias, self.linear2.weight,
                self.linear2.bias, self.norm1.weight, self.norm1.bias, self.norm2.weight, self.norm2.bias]


class _EncoderWeights(nn.Module):
    def __init__(self, e: int, ffn: int, layers: int):
        super().__init__()
        self.layers = nn.ModuleList([_EncoderLayerWeights(e, ffn) for _ in range(layers)])


class TransformerLstm(AbstractCaterModel):
    """reference learned_models.py:154-197.  The reference hands [B*T, 15, E] to a sequence-first encoder:
    attention spans the S = B*T frame axis (all clips of the minibatch, non-causal) independently per slot, and
    only slot 0 is kept - so only slot 0 is evaluated here (the same function, and the same gradients: slots 1..14
    never reach the loss).  dim_feedforward is torch's default 2048.

    Training: nn.TransformerEncoderLayer's dropout (default 0.1, active under model.train()) is drawn from a
    counter-based generator (`dropout_seed`, advanced every forward) - the reference's torch masks cannot be
    reproduced, so gradient parity is pinned with `dropout = 0.0`."""

    FFN = 2048
    dropout = 0.1
    dropout_seed = 0x5EED

    def __init__(self, config: Dict[str, int]):
        super().__init__(config)
        e = config["boxes_features_dim"]
        self._e, self._nhead = e, config["num_attention_heads"]
        self._nl = config["num_attention_layers"]
        h, ll = config["lstm_hidden_dim"], config["num_lstm_layers"]
        self.boxes_linear = LinearWeight(self.bb_in_dim, e)
        self.attention_encoder = _EncoderWeights(e, self.FFN, self._nl)
        self.video_LSTM = LSTMWeights(e, h, num_layers=ll)
        self.predictions_layer = LinearWeight(h, self.bb_out_dim)
        self._runner = _LstmStackRunner(ll, e, h)
        self._ews: Dict[int, torch.Tensor] = {}      # encoder workspace per stream
        self._tscratch = None
        self._calls = 0
        # tests only: {layer: (attention [nhead, S, S], after out_proj [S, E], after ReLU [S, ffn], after linear2 [S, E])} uint8
        # device tensors, nonzero = keep - the NEXT training forward draws its dropout from them instead of the generator
        self._test_dropout_masks = None

    def _forward_train(self, x: torch.Tensor) -> torch.Tensor:
        B, T = int(x.shape[0]), int(x.shape[1])
        p = float(self.dropout) if self.training else 0.0
        self._calls += 1
        z = _SlotEmbedFunction.apply(x, self.boxes_linear.weight, 1).view(B * T, self._e)
        for li, layer in enumerate(self.attention_encoder.layers):
            ts = layer.tensors()
            for t_ in ts:
                if t_.device != x.device or not t_.is_contiguous() or t_.dtype != torch.float32:
                    raise RuntimeError("parameters must be contiguous fp32 on the input's device")
            seed = (int(self.dropout_seed) + 1000003 * self._calls + 7919 * li) & 0xFFFFFFFFFFFF
            if self._test_dropout_masks is not None:        # tests: this layer's four masks from buffers (the reference's draws)
                m = self._test_dropout_masks[li]
                _lib.check(_lib.load().opseq_encoder_test_masks_set(li, seed, *(t_.data_ptr() for t_ in m)),
                           "opseq_encoder_test_masks_set")
            z = _EncoderLayerTrainFunction.apply(z, self, p, seed, *ts)
        return self._runner.run_train(z.view(B, T, self._e), self.video_LSTM, self.predictions_layer)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        _check_input(self, x, 5)
        x = x.contiguous().float()
        if _wants_grad(self):
            with torch.cuda.device(x.device):
                return self._forward_train(x)
        return self._forward_eval(x, 1)

    # tokens one merged pass may carry: the FFN activation [tokens][2048] fp32 has to stay below 2 GiB (include/opnet_hip.h)
    MAX_TOKENS_PER_PASS = 192 * 1024

    def forward_segments(self, x: torch.Tensor, n_seg: int, exact: bool = False) -> torch.Tensor:
        """x [n_seg * b, T, 15, 5] = n_seg INDEPENDENT requests of b clips each, back to back -> y [n_seg * b, T, 4].  The reference
        serves a request per call (learned_models.py:176-197: attention over S = b * T, the clips OF THAT CALL); here the token-wise
        stages (embedding, in / out projection, FFN, layer norms) run over all requests' tokens at once, attention stays inside a
        request and ONE persistent launch runs the stacked LSTM over all clips - a one-clip request alone leaves 6 of 8 XCDs and 3
        of 4 MFMA columns of that launch idle.  Two forms:
          * a pass below _LstmStackRunner.XCDT_MIN_BATCH clips, or exact = True: every kernel is the one the LONE request would run
            (opseq_encoder_layer_segmented_f32, the 4-clip persistent stack): every request's rows are bit-identical to
            `forward(request)` alone;
          * otherwise the throughput form: token-wise products on large tiles chosen by all rows
            (opseq_encoder_layer_batched_f32), the stacked LSTM on 16-clip groups (csrc/seq_xcdt_kernels.hip): each request agrees
            with its lone forward to rounding (other summation orders; both are held to the reference goldens).
        Inference only."""
        _check_input(self, x, 5)
        if n_seg <= 0 or int(x.shape[0]) % n_seg:
            raise ValueError(f"{int(x.shape[0])} clips do not split into {n_seg} equal requests")
        if _wants_grad(self):
            raise RuntimeError("forward_segments is the serving path: call it under torch.no_grad() / after eval()")
        return self._forward_eval(x.contiguous().float(), int(n_seg), bool(exact))

    def max_requests_per_pass(self, b: int, T: int, exact: bool = False) -> int:
        """how many requests of b clips x T frames `forward_segments` takes in one pass: the token limit above; exact: the stacked
        LSTM must run on the SAME engine as for the lone request (the 4-clip persistent launch carries at most opseq_xcd_max_batch
        clips; the launch chain sums in another order); otherwise one launch of the throughput form (opseq_xcdt_max_batch clips)"""
        n_tok = max(1, self.MAX_TOKENS_PER_PASS // max(b * T, 1))
        r = self._runner
        lib = _lib.load()
        n_exact = n_tok
        if r.engine(b, T) == "x":
            n_exact = min(n_tok, max(1, int(lib.opseq_xcd_max_batch(r.L)) // b))
        if exact or not r._wants_xcdt(max(n_tok * b, r.XCDT_MIN_BATCH), T):
            return n_exact
        n_t = min(n_tok, max(1, int(lib.opseq_xcdt_max_batch(T, r.L, r.KX, r.H)) // b))
        return n_t if n_t * b >= r.XCDT_MIN_BATCH and n_t > n_exact else n_exact

    def pass_engine(self, n_seg: int, b: int, T: int, exact: bool = False) -> str:
        """the stacked LSTM's engine for a pass of n_seg requests of b clips x T frames ("t" / "x" / "c", _LstmStackRunner.engine):
        the lone request's when exact (or alone), else the whole pass's"""
        lone = self._runner.engine(b, T)
        return lone if (exact or n_seg == 1) else self._runner.engine(n_seg * b, T)

    def _forward_eval(self, x: torch.Tensor, n_seg: int, exact: bool = False) -> torch.Tensor:
        if n_seg > 1 and n_seg > self.max_requests_per_pass(int(x.shape[0]) // n_seg, int(x.shape[1]), exact):
            raise ValueError(f"{n_seg} requests of {int(x.shape[0]) // n_seg} clips x {int(x.shape[1])} frames exceed one pass "
                             f"(max_requests_per_pass = {self.max_requests_per_pass(int(x.shape[0]) // n_seg, int(x.shape[1]), exact)})")
        if self.training and self.dropout > 0:
            raise RuntimeError("TransformerLstm: train mode without gradients would still apply dropout; call eval() "
                               "for inference")
        lib = _lib.load()
        B, T = int(x.shape[0]), int(x.shape[1])
        St, e, dev = B * T, self._e, x.device
        S = St // n_seg                          # tokens of one request: what attention spans (exact form: and every kernel is chosen by)
        # the stacked LSTM's engine: the lone request's (every request then comes out as it would alone), or - a pass of at least
        # XCDT_MIN_BATCH clips that need not be bit-identical - the throughput form
        lone = self._runner.engine(B // n_seg, T)
        engine = lone if (exact or n_seg == 1) else self._runner.engine(B, T)
        if engine != "t":
            engine = lone
        self.last_pass_engine = engine
        if n_seg > 1 and St > self.MAX_TOKENS_PER_PASS:
            raise ValueError(f"{St} tokens in one pass (limit {self.MAX_TOKENS_PER_PASS}): split the requests")
        with torch.cuda.device(dev):
            stream = _stream_ptr(dev)
            z = torch.empty((St, e), dtype=torch.float32, device=dev)
            rc = lib.opseq_slot_embed_relu_f32(x.data_ptr(), self.boxes_linear.weight.data_ptr(), z.data_ptr(), St, 1, e, stream)
            _lib.check(rc, "opseq_slot_embed_relu_f32")
            nb = lib.opseq_encoder_workspace_bytes(St, e, self._nhead, self.FFN)
            if nb == 0:
                _lib.check(-2, "opseq_encoder_workspace_bytes")
            # one encoder workspace per stream: a server keeps two passes in flight on two streams (serving.py)
            ews = self._ews.get(stream)
            if ews is None or ews.numel() < nb or ews.device != dev:
                if len(self._ews) >= 4:
                    self._ews.pop(next(iter(self._ews)))
                ews = self._ews[stream] = torch.empty(nb, dtype=torch.uint8, device=dev)
            for layer in self.attention_encoder.layers:
                ts = layer.tensors()
                for t_ in ts:
                    if t_.device != dev or not t_.is_contiguous() or t_.dtype != torch.float32:
                        raise RuntimeError("parameters must be contiguous fp32 on the input's device")
                if n_seg == 1:
                    rc = lib.opseq_encoder_layer_f32(z.data_ptr(), *(t_.data_ptr() for t_ in ts), ews.data_ptr(),
                                                     ews.numel(), S, e, self._nhead, self.FFN, stream)
                    _lib.check(rc, "opseq_encoder_layer_f32")
                else:
                    fn = lib.opseq_encoder_layer_batched_f32 if engine == "t" and lone != "t" else lib.opseq_encoder_layer_segmented_f32
                    rc = fn(z.data_ptr(), *(t_.data_ptr() for t_ in ts), ews.data_ptr(), ews.numel(), S, n_seg, e, self._nhead,
                            self.FFN, stream)
                    _lib.check(rc, "opseq_encoder_layer_segmented_f32")
            return self._runner.run(z.view(B, T, e), self.video_LSTM, self.predictions_layer, engine=engine)


class OPNetLstmMlp(AbstractCaterModel):
    """reference learned_models.py:55-89: OPNet with relu(Linear 6->H2) in place of the video LSTM."""

    def __init__(self, config: Dict[str, int]):
        super().__init__(config)
        self.bb_in_dim = 6
        if config["object_to_track_pred_dim"] != 15:
            raise ValueError("object_to_track_pred_dim must be 15 (number of object slots)")
        h1, h2 = config["object_to_track_hidden_dim"], config["videos_hidden_dim"]
        self.object_to_track_LSTM = LSTMWeights(self.bb_in_dim * 15, h1)
        self.object_to_track_prediction = LinearWeight(h1, 15)
        self.hidden_layer = LinearWeight(self.bb_in_dim, h2)
        self.prediction_layer = LinearWeight(h2, self.bb_out_dim)
        self._h1, self._h2 = h1, h2
        self._packed, self._key, self._ws = None, None, {}
        self._tpacked, self._tscratch, self._tws, self._tws_key, self._train_gen = None, None, None, None, 0

    def forward(self, boxes: torch.Tensor):
        _check_input(self, boxes, 6)
        lib = _lib.load()
        boxes = boxes.contiguous().float()
        B, T, dev = int(boxes.shape[0]), int(boxes.shape[1]), boxes.device
        ws_list = [self.object_to_track_LSTM.weight_ih_l0, self.object_to_track_LSTM.weight_hh_l0,
                   self.object_to_track_prediction.weight, self.hidden_layer.weight, self.prediction_layer.weight]
        if _wants_grad(self):
            for w in ws_list:
                if w.device != dev or w.dtype != torch.float32 or not w.is_contiguous():
                    raise RuntimeError("OPNetLstmMlp parameters must be contiguous fp32 on the input's device")
            return _OPNetMlpTrainFunction.apply(self, boxes, *ws_list)
        with torch.cuda.device(dev):
            stream = _stream_ptr(dev)
            key = _weights_key(ws_list, dev)
            if self._key != key:
                nbytes = lib.opnet_packed_weights_bytes(self._h1, self._h2)
                if nbytes == 0:
                    _lib.check(-2, "opnet_packed_weights_bytes")
                if self._packed is None or self._packed.device != dev:
                    self._packed = torch.zeros(nbytes // 4, dtype=torch.float32, device=dev)
                rc = lib.opnet_mlp_pack_weights_f32(*(w.data_ptr() for w in ws_list), self._packed.data_ptr(), nbytes,
                                                    self._h1, self._h2, stream)
                _lib.check(rc, "opnet_mlp_pack_weights_f32")
                self._key = key
            wkey = (B, T, str(dev), stream)
            if wkey not in self._ws:
                self._ws = {wkey: torch.empty(lib.opnet_workspace_bytes(B, T, self._h1, self._h2), dtype=torch.uint8, device=dev)}
            ws = self._ws[wkey]
            y = torch.empty((B, T, 4), dtype=torch.float32, device=dev)
            logits = torch.empty((B, 15, T), dtype=torch.float32, device=dev)
            rc = lib.opnet_mlp_forward_f32(boxes.data_ptr(), self._packed.data_ptr(), y.data_ptr(), logits.data_ptr(),
                                           ws.data_ptr(), ws.numel(), B, T, self._h1, self._h2, stream)
            _lib.check(rc, "opnet_mlp_forward_f32")
        return y, logits
