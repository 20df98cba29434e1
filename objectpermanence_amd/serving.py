"""Request batching in front of the reasoners.

The reference calls its model one DataLoader minibatch at a time (baselines/inference_main.py:191-217, batch_size 16 in
configs/inference_config.json).  On the MI355X the unit that fills the chip is much larger: the per-XCD persistent OPNet
forward (csrc/opnet_xcd_kernels.hip) wants at least two 16-clip groups on each of the 8 XCDs, i.e. >= 256 clips per launch,
and gives 99 k clips/s there against 26 k for one 32-clip batch alone.  Clips are independent
(SURVEY.md 8-e1), so concurrent requests can simply be concatenated: `ReasonerServer` collects submitted minibatches,
runs them as ONE forward when `max_clips` are pending (or on `flush()`), and hands every request its own slice of the
outputs.  Results are bit-identical to running the concatenation through `model(...)` directly.

TransformerLstm's attention spans all clips of a minibatch: concatenating requests would change its
results.  Its requests are therefore merged as SEGMENTS (`TransformerLstm.forward_segments`): every submitted minibatch stays
one sequence that attends only to itself, the token-wise stages of all pending requests run as one set of launches and one
persistent launch runs the stacked LSTM over all their clips - each request's result is bit-identical to `model(request)` alone
(config 3: one clip per request; 16 of them cost about what one costs).  Requests of one pass must have the same shape [b, T].
A pass is ~0.55 ms of encoder launches followed by the 0.67 ms persistent stacked-LSTM launch, which leaves the matrix pipes 83 %
idle: such a server therefore issues its passes on TWO side streams in turn, so that one pass's encoder runs under the other's
recurrence (persistent launches themselves are serialised per device inside the library): 13.8 k -> 17.4 k clips/s, same bits.
A pass in the THROUGHPUT form (>= 64 clips: the 16-clip persistent launch at 0.8 matrix-pipe busy, large GEMM tiles) has no such
idle time to fill - a second pass's GEMM workgroups on the same CUs only slow the recurrence's critical path: measured 25.7 k
clips/s with two passes in flight against 28.2 k one after the other at 256 requests per pass - so those passes share ONE side stream.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

from .supported_models import DOUBLE_OUTPUT_MODELS


class PendingResult:
    """Handle returned by ReasonerServer.submit(); `.result()` flushes the server if the request is still queued."""

    def __init__(self, server: "ReasonerServer", n_clips: int):
        self._server, self.n_clips = server, n_clips
        self._value = None
        self._error = None
        self._event = None           # set when the forward ran on a side stream: the consumer's stream waits for it in result()

    def done(self) -> bool:
        return self._value is not None or self._error is not None

    def result(self):
        if self._value is None and self._error is None:
            self._server.flush()
        if self._error is not None:      # the forward this request was part of failed: every request of it says so
            raise self._error
        if self._event is not None:      # (no host wait: the stream the caller works on is made to wait for the pass)
            cur = torch.cuda.current_stream(self._server._device)
            cur.wait_event(self._event)
            for t in (self._value if isinstance(self._value, tuple) else (self._value,)):
                t.record_stream(cur)
            self._event = None
        return self._value


class ReasonerServer:
    def __init__(self, model: torch.nn.Module, model_name: str = "opnet", max_clips: int = 1024, concat: bool = True,
                 streams: Optional[int] = None, exact: bool = False):
        # a model whose clips are coupled inside a request (TransformerLstm) merges requests as segments, never by concatenation
        self.segmented = hasattr(model, "forward_segments")
        # passes in flight: 2 for a segmented model on a GPU (module docstring), else the forward runs on the caller's stream
        self._n_streams = int(streams) if streams is not None else (2 if self.segmented else 1)
        self._side: List["torch.cuda.Stream"] = []
        self._device = None
        self.model, self.model_name, self.max_clips = model, model_name, int(max_clips)
        # concat = False: OPNet's persistent launch reads the request tensors where they lie (OPNet.forward_requests) instead
        # of one torch.cat - it saves the 108 KB/clip copy, but the pack kernel then walks up to 64 sources and the host
        # builds the pointer table: measured 118-122 k clips/s against 120-124 k for twenty 32-clip requests, so the copy stays
        # the default
        self.concat = bool(concat)
        # segmented models: exact = True keeps every pass on the kernels a lone request runs (each result bit-identical to
        # `model(request)`); False lets a pass of _LstmStackRunner.XCDT_MIN_BATCH clips or more take the throughput form (16-clip column groups, large GEMM tiles:
        # agrees with the lone forward to rounding) and carry up to opseq_xcdt_max_batch clips
        self.exact = bool(exact)
        self._queue: List[Tuple[torch.Tensor, PendingResult]] = []
        self._pending = 0
        self._pass_limit = {}        # (b, T) of a request -> clips one pass takes (segmented models)
        # optional callable run on the host right before the forward is enqueued (after the requests were concatenated): a
        # data-parallel caller makes the launch wait for its previous collective here - a persistent launch needs every CU of
        # the device, and an RCCL kernel that holds a few of them while it waits for a slower rank would stall it
        self.before_launch = None
        self.last_output = None      # the whole output of the last forward (callers that post-process per launch)
        self.forwards = 0            # statistics: forwards issued / clips served
        self.clips = 0

    def submit(self, boxes: torch.Tensor) -> PendingResult:
        """boxes [b, T, 15, F] on the model's device.  Returns a handle; the forward runs when `max_clips` clips are
        pending or on flush() / handle.result()."""
        if self._queue:
            first = self._queue[0][0]
            if boxes.device != first.device or boxes.dtype != first.dtype:
                raise ValueError(f"request on {boxes.device} / {boxes.dtype} while {first.device} / {first.dtype} requests are "
                                 "pending: requests of one server share a launch and must share device and dtype")
            if tuple(boxes.shape[1:]) != tuple(first.shape[1:]) or (self.segmented and boxes.shape[0] != first.shape[0]):
                self.flush()         # a different clip length (segments: a different request shape) cannot share a launch
        h = PendingResult(self, int(boxes.shape[0]))
        self._queue.append((boxes, h))
        self._pending += h.n_clips
        limit = self.max_clips
        if self.segmented:           # as many requests as one pass takes (exact: and returns bit-identical to their lone forwards)
            key = (int(boxes.shape[0]), int(boxes.shape[1]))
            per_pass = self._pass_limit.get(key)
            if per_pass is None:
                per_pass = self._pass_limit[key] = self.model.max_requests_per_pass(key[0], key[1], self.exact) * key[0]
            limit = min(limit, per_pass)
        if self._pending >= limit:
            self.flush()
        return h

    def _side_stream(self, device: torch.device, queue=None):
        """the side stream of the next pass (round robin; always the first one for a pass in the throughput form: module docstring),
        or None: one stream asked for / not a GPU"""
        if self._n_streams <= 1 or device.type != "cuda":
            return None
        if self._device != device:
            self._device, self._side = device, [torch.cuda.Stream(device=device) for _ in range(self._n_streams)]
        if self.segmented and queue and not self.exact and hasattr(self.model, "pass_engine"):
            b, T = int(queue[0][0].shape[0]), int(queue[0][0].shape[1])
            if self.model.pass_engine(len(queue), b, T, False) == "t":
                return self._side[0]
        return self._side[self.forwards % len(self._side)]

    def _forward(self, queue):
        if self.segmented:
            x = queue[0][0] if len(queue) == 1 else torch.cat([q[0] for q in queue], dim=0)
            if self.before_launch is not None:
                self.before_launch()
            return self.model(x) if len(queue) == 1 else self.model.forward_segments(x, len(queue), self.exact)
        if len(queue) > 1 and not self.concat and hasattr(self.model, "forward_requests"):
            if self.before_launch is not None:
                self.before_launch()
            return self.model.forward_requests([q[0] for q in queue])     # OPNet: one launch over the requests where they lie
        x = queue[0][0] if len(queue) == 1 else torch.cat([q[0] for q in queue], dim=0)
        if self.before_launch is not None:
            self.before_launch()
        return self.model(x)

    @torch.no_grad()
    def flush(self) -> None:
        if not self._queue:
            return
        queue, self._queue, self._pending = self._queue, [], 0
        event = None
        try:
            side = self._side_stream(queue[0][0].device, queue)
            if side is None:
                out = self._forward(queue)
            else:
                side.wait_stream(torch.cuda.current_stream(self._device))      # the requests were produced on the caller's stream
                with torch.cuda.stream(side):
                    for boxes, _ in queue:
                        boxes.record_stream(side)
                    out = self._forward(queue)
                    event = torch.cuda.Event()
                    event.record(side)
        except Exception as e:
            # a bad shape, an out-of-memory concatenation, an ABI error: the requests of this forward are not silently lost -
            # each handle re-raises from result() (and done() turns true), the server stays usable
            for _, h in queue:
                h._error = e
            raise
        double = isinstance(out, tuple)
        self.last_output = out if double else (out,)
        self.forwards += 1
        self.clips += sum(h.n_clips for _, h in queue)
        lo = 0
        for boxes, h in queue:
            hi = lo + h.n_clips
            h._value = (out[0][lo:hi], out[1][lo:hi]) if double else out[lo:hi]
            h._event = event
            lo = hi

    def infer(self, boxes: torch.Tensor):
        """submit + wait: what a caller without concurrency gets (one forward per call)"""
        h = self.submit(boxes)
        return h.result()


def output_boxes(model_name: str, out):
    """the y_boxes part of a model output (DOUBLE_OUTPUT_MODELS return (y_boxes, logits), inference_main.py:203-207)"""
    return out[0] if model_name in DOUBLE_OUTPUT_MODELS else out
