"""Loss and optimiser of the reference's training step on device.

* ``l1_mean(y, labels)``  == ``torch.mean(nn.L1Loss(reduction="none")(y, labels))``
  (reference baselines/training_main.py:152,192,204) - one fused forward+backward kernel pair.
* ``FusedAdam``           == ``torch.optim.Adam(params, lr)`` with the reference's defaults
  (training_main.py:150): betas (0.9, 0.999), eps 1e-8, no weight decay.  It is a
  ``torch.optim.Optimizer`` so ``ReduceLROnPlateau`` (training_main.py:151,247) drives it unchanged.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib


class _L1Mean(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y, labels, beta=0.0):
        if not y.is_cuda:
            raise RuntimeError("l1_mean runs on the GPU only (no CPU fallback)")
        lib = _lib.load()
        y = y.contiguous().float()
        labels = labels.to(y.device).contiguous().float()
        n = y.numel()
        loss = torch.empty((), dtype=torch.float32, device=y.device)
        dy = torch.empty_like(y)
        scratch = torch.empty(4096, dtype=torch.uint8, device=y.device)
        with torch.cuda.device(y.device):
            st = torch.cuda.current_stream(y.device).cuda_stream
            if beta > 0:
                rc = lib.opnet_smooth_l1_loss_f32(y.data_ptr(), labels.data_ptr(), loss.data_ptr(), dy.data_ptr(), n,
                                                  float(beta), scratch.data_ptr(), scratch.numel(), st)
            else:
                rc = lib.opnet_l1_loss_f32(y.data_ptr(), labels.data_ptr(), loss.data_ptr(), dy.data_ptr(), n,
                                           scratch.data_ptr(), scratch.numel(), st)
        _lib.check(rc, "opnet_l1_loss_f32")
        ctx.save_for_backward(dy)
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        (dy,) = ctx.saved_tensors
        return dy * grad_out, None, None


def loss_and_grad(y: torch.Tensor, labels: torch.Tensor, beta: float = 0.0):
    """(loss, dloss/dy) of the supervised loss WITHOUT an autograd node: a training step that calls `y.backward(dy)` itself
    saves the two launches `loss.backward()` spends on the implicit gradient 1.0 (a fill) and on `dy * 1.0`."""
    if not y.is_cuda:
        raise RuntimeError("loss_and_grad runs on the GPU only (no CPU fallback)")
    lib = _lib.load()
    yd = y.detach().contiguous().float()
    labels = labels.to(yd.device).contiguous().float()
    n = yd.numel()
    loss = torch.empty((), dtype=torch.float32, device=yd.device)
    dy = torch.empty_like(yd)
    scratch = torch.empty(4096, dtype=torch.uint8, device=yd.device)
    with torch.cuda.device(yd.device):
        st = torch.cuda.current_stream(yd.device).cuda_stream
        if beta > 0:
            rc = lib.opnet_smooth_l1_loss_f32(yd.data_ptr(), labels.data_ptr(), loss.data_ptr(), dy.data_ptr(), n, float(beta),
                                              scratch.data_ptr(), scratch.numel(), st)
        else:
            rc = lib.opnet_l1_loss_f32(yd.data_ptr(), labels.data_ptr(), loss.data_ptr(), dy.data_ptr(), n, scratch.data_ptr(),
                                       scratch.numel(), st)
    _lib.check(rc, "opnet_l1_loss_f32")
    return loss, dy


def l1_mean(y: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
    return _L1Mean.apply(y, labels, 0.0)


def smooth_l1_mean(y: torch.Tensor, labels: torch.Tensor, beta: float = 1.0) -> torch.Tensor:
    """torch.nn.SmoothL1Loss(beta=beta)(y, labels) - the loss BASELINE.json's config 2 names (the reference
    trains with L1; this is the extra knob SURVEY.md section 0 asks for)."""
    return _L1Mean.apply(y, labels, float(beta))


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, grad_scale: float = 1.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, grad_scale=grad_scale))
        # device-side guards of the next step() (opnet_adam_multi_step_guarded_f32; set by training.train_step, None = off):
        # abort word of the persistent launches behind the gradients | the loss (skip if not finite) | data-parallel guard
        self.abort_ptr = None
        self.loss_ptr = None
        self.guard_ptr = None
        self._last_stepped = []

    def rollback_step_count(self) -> None:
        """undo the step COUNTERS of the last step(): for a step whose update the device-side guard skipped (parameters and
        moments untouched) and that the caller is about to repeat"""
        for st in self._last_stepped:
            st["step"] -= 1
        self._last_stepped = []

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        self._last_stepped = []
        for group in self.param_groups:
            b1, b2 = group["betas"]
            # parameters of one device that are at the same step share ONE launch (opnet_adam_multi_step_f32: up to 16 tensors)
            batches = {}
            keep = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("FusedAdam runs on the GPU only (no CPU fallback)")
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                self._last_stepped.append(st)
                g = p.grad.contiguous()
                keep.append(g)
                batches.setdefault((p.device, int(st["step"])), []).append((p, g, st))
            for (dev, step_no), items in batches.items():
                with torch.cuda.device(dev):
                    stream = torch.cuda.current_stream(dev).cuda_stream
                    for lo in range(0, len(items), 16):
                        chunk = items[lo:lo + 16]
                        n = len(chunk)
                        arr = lambda vals: (ctypes.c_void_p * n)(*vals)
                        rc = lib.opnet_adam_multi_step_guarded_f32(
                            n, arr([p.data_ptr() for p, _, _ in chunk]), arr([g.data_ptr() for _, g, _ in chunk]),
                            arr([st["exp_avg"].data_ptr() for _, _, st in chunk]),
                            arr([st["exp_avg_sq"].data_ptr() for _, _, st in chunk]),
                            (ctypes.c_long * n)(*[p.numel() for p, _, _ in chunk]), float(group["lr"]), float(b1), float(b2),
                            float(group["eps"]), step_no, float(group["grad_scale"]), self.abort_ptr, self.loss_ptr,
                            self.guard_ptr, stream)
                        _lib.check(rc, "opnet_adam_multi_step_guarded_f32")
                # the update happened outside torch's view: bump the version counters so that consumers keyed on them
                # (the modules' packed-weight caches) see the parameters as modified
                for p, _, _ in items:
                    torch.autograd.graph.increment_version(p)
        return loss
