"""One training step of a reasoner, single- or multi-GPU, with the reference's semantics
(reference baselines/training_main.py:183-217): zero_grad -> forward -> loss -> backward -> Adam.

Loss selection mirrors training_main.py:192-210: supervised models use mean(|y - label|);
``*_no_labels`` models use mean(|y - label| * mask) + 0.5 * mean(||y[:,1:] - y[:,:-1]||_2).  The latter is
a few elementwise torch ops on [B,T,4] whose gradient enters the HIP backward through dL/dy.

Data parallel (not in the reference - SURVEY.md section 2.3): every rank runs the step on its own clips,
then ONE all-reduce over the flat 5.68 MB gradient buffer (parallel.all_reduce_gradients) on a side
stream, weighted n_local/n_global so the result equals the single-process mean-loss gradient.
"""
from __future__ import annotations

import math
import warnings
from typing import Optional

import torch
import torch.distributed as dist

from . import parallel
from .optim import l1_mean, loss_and_grad, smooth_l1_mean
from .supported_models import DOUBLE_OUTPUT_MODELS, NO_LABELS_MODELS


def _consistency(output: torch.Tensor) -> torch.Tensor:
    nxt, cur = output[:, 1:, :], output[:, :-1, :]
    return torch.mean(torch.norm(nxt - cur, p=2, dim=-1)) if output.shape[1] > 1 else output.new_zeros(())


def compute_loss(model_name: str, output: torch.Tensor, labels: torch.Tensor, mask: Optional[torch.Tensor] = None,
                 loss_kind: str = "l1", with_consistency: bool = True):
    """Returns (loss, pred_loss, consistency_loss) as training_main.py:192-210 does.  loss_kind "smooth_l1" swaps the
    supervised L1 for torch.nn.SmoothL1Loss (BASELINE.json config 2 names it; the reference itself trains with L1).
    The supervised models' loss does not contain the consistency term (the reference only PRINTS it, :213-214); a caller that
    does not print it passes with_consistency=False and gets None - four launches less per training step."""
    if model_name in NO_LABELS_MODELS:
        consistency = _consistency(output)
        pred = torch.mean(torch.abs(output - labels) * mask)
        return pred + 0.5 * consistency, pred, consistency
    pred = smooth_l1_mean(output, labels) if loss_kind == "smooth_l1" else l1_mean(output, labels)
    return pred, pred, (_consistency(output.detach()) if with_consistency else None)


def train_step(model_name: str, model: torch.nn.Module, optimizer: torch.optim.Optimizer, boxes: Optional[torch.Tensor],
               labels: Optional[torch.Tensor], mask: Optional[torch.Tensor] = None, group: Optional[dist.ProcessGroup] = None,
               n_global: Optional[int] = None, comm_stream: Optional[torch.cuda.Stream] = None,
               loss_kind: str = "l1", overlap=None, comm_events: Optional[list] = None) -> torch.Tensor:
    """zero_grad -> forward -> loss -> backward -> [gradient all-reduce] -> Adam (training_main.py:183-217).

    Data parallel: EVERY rank calls this for every global batch - a rank whose slice of the batch is empty passes
    boxes=None, contributes zero gradients with weight 0 / n_global and still joins the collective and the optimiser step
    (otherwise the other ranks would wait in the all-reduce forever and this rank's weights would drift).  The gradients
    live in one flat bucket (parallel.GradBucket) that the HIP backward writes in place; the all-reduce is enqueued on
    `comm_stream` as soon as the backward (its last kernel is the merged weight-gradient GEMM) is enqueued, `overlap()` -
    the caller's work that does not depend on the new weights: the next batch's host-to-device copies, its input packing -
    runs on the current stream meanwhile, and only then does the current stream wait for the collective and run Adam."""
    distributed = parallel.is_active(group)       # > 1 rank, or a forced group of one (OPNET_FORCE_DIST=1: the GPU tests)
    bucket = getattr(model, "_grad_bucket", None)
    if bucket is None:
        bucket = model._grad_bucket = parallel.GradBucket(model.parameters())
    optimizer.zero_grad(set_to_none=True)
    n_local = 0 if boxes is None else int(boxes.shape[0])
    guard_word = None
    if n_local > 0:
        out = model(boxes)
        output = out[0] if model_name in DOUBLE_OUTPUT_MODELS else out
        if model_name not in NO_LABELS_MODELS and output.is_cuda:
            # supervised loss: value and gradient from one call, the backward started from dy (no autograd node for the loss)
            loss, dy = loss_and_grad(output, labels, 1.0 if loss_kind == "smooth_l1" else 0.0)
            output.backward(dy)
        else:
            loss, _, _ = compute_loss(model_name, output, labels, mask, loss_kind, with_consistency=False)
            loss.backward()
            loss = loss.detach()
        guard_word = model.launch_guard() if hasattr(model, "launch_guard") else None
    else:
        loss = bucket.flat.new_zeros(())
    if distributed or n_local == 0:
        bucket.collect(fill_missing=distributed)       # (one process: the optimiser reads the gradients where autograd left them -
                                                       # 37 copies into the bucket a step for transformer_lstm otherwise)
    # device-side guards of the optimiser step (optim.FusedAdam): an aborted persistent launch (its gradients are NaN) or a
    # non-finite loss must not reach the weights; the host learns of it at its next sync point (step_aborted below)
    abort_ptr = guard_word.data_ptr() if guard_word is not None else None
    loss_ptr = loss.data_ptr() if (loss.is_cuda and n_local > 0) else None
    on_device_dp = distributed and bucket.flat.is_cuda
    n_glob = (n_global if n_global is not None else n_local * dist.get_world_size(group)) if distributed else n_local
    if on_device_dp:
        # this rank's abort word and "my loss is not finite" -> the guard slots the all-reduce sums over the ranks, so that
        # every rank's optimiser takes the SAME decision (a rank-local skip would let the replicas drift apart)
        from . import _lib
        with torch.cuda.device(bucket.flat.device):
            _lib.check(_lib.load().opnet_dp_guard_f32(bucket.guard.data_ptr(), abort_ptr, loss_ptr,
                                                      float(n_local) / float(max(n_glob, 1)), torch.cuda.current_stream(bucket.flat.device).cuda_stream),
                       "opnet_dp_guard_f32")
        abort_ptr = loss_ptr = None         # both travel through the guard slots now
    if hasattr(optimizer, "abort_ptr"):
        optimizer.abort_ptr, optimizer.loss_ptr = abort_ptr, loss_ptr
        optimizer.guard_ptr = bucket.guard.data_ptr() if on_device_dp else None
    if distributed:
        cur = torch.cuda.current_stream(bucket.flat.device) if bucket.flat.is_cuda else None
        st = comm_stream or cur
        if cur is not None:
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                if comm_events is not None:     # measurement (bench.py): the collective's own time on its stream
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(st)
                bucket.all_reduce(n_local, max(n_glob, 1), group)
                if comm_events is not None:
                    e1.record(st)
                    comm_events.append((e0, e1))
            if overlap is not None:
                overlap()
            cur.wait_stream(st)
        else:                                   # CPU tensors (gloo tests)
            bucket.all_reduce(n_local, max(n_glob, 1), group)
            if overlap is not None:
                overlap()
    elif overlap is not None:
        overlap()
    try:
        optimizer.step()
    finally:
        # the guards are raw device addresses of THIS step's tensors (loss, workspace, bucket): a later optimizer.step() outside
        # train_step must not read memory that has been freed or reused since
        if hasattr(optimizer, "abort_ptr"):
            optimizer.abort_ptr = optimizer.loss_ptr = optimizer.guard_ptr = None
    return loss


def global_loss(model: torch.nn.Module, loss: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """the loss of the WHOLE minibatch of the train_step that returned `loss` (what the reference prints per step,
    training_main.py:212-214): data parallel on device it is the all-reduced guard slot 2 - sum over ranks of loss_r * n_r / n,
    carried by the gradient all-reduce - otherwise `loss` itself"""
    bucket = getattr(model, "_grad_bucket", None)
    if bucket is not None and parallel.is_active(group) and bucket.flat.is_cuda:
        return bucket.guard[2]
    return loss


def step_aborted(model: torch.nn.Module, group: Optional[dist.ProcessGroup] = None) -> bool:
    """Call after the host has synchronised with a train_step (the driver's float(loss)): did a persistent launch of that
    step give up on any rank?  If so the guarded optimiser left parameters and moments untouched on every rank, this process
    has switched to the launch chain, and the caller repeats the step: optimizer.rollback_step_count(); train_step(...)."""
    fn = getattr(model, "training_step_aborted", None)
    bad = bool(fn()) if fn is not None else False
    bucket = getattr(model, "_grad_bucket", None)
    if bucket is not None and parallel.is_active(group) and bucket.flat.is_cuda and float(bucket.guard[0]) != 0.0:
        bad = True
        from . import _lib              # some OTHER rank's launch gave up: every rank leaves the persistent kernels together
        _lib.load().opnet_xcd4_enable(0)
        _lib.load().opseq_xcd_enable(0)
    return bad


_warned_nonfinite = False


def step_skipped_nonfinite(model: torch.nn.Module, optimizer, loss_value: float,
                           group: Optional[dist.ProcessGroup] = None) -> bool:
    """Call after the driver's float(loss), next to step_aborted: was the update of that step skipped because a loss was not
    finite?  The guarded Adam then left parameters and moments untouched (torch.optim.Adam, training_main.py:217, would have
    written NaN into every weight); here the step COUNTERS are rolled back too - the bias correction must not advance over an
    update that never happened - and the first occurrence warns.  Data parallel: the decision was taken from the all-reduced
    guard slot, so every rank skipped together and every rank reports it, whichever rank's loss it was."""
    global _warned_nonfinite
    skipped = not math.isfinite(loss_value)
    bucket = getattr(model, "_grad_bucket", None)
    if bucket is not None and parallel.is_active(group) and bucket.flat.is_cuda and float(bucket.guard[1]) != 0.0:
        skipped = True
    if skipped:
        if hasattr(optimizer, "rollback_step_count"):
            optimizer.rollback_step_count()
        if not _warned_nonfinite:
            _warned_nonfinite = True
            warnings.warn("objectpermanence_amd: a training step's loss is not finite; its optimiser update was skipped on every "
                          "rank (weights, moments and step count unchanged)", RuntimeWarning, stacklevel=2)
    return skipped
