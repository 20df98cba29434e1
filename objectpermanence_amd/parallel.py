"""Video-level data parallelism for the reasoners: one process per GPU, clips sharded in contiguous
blocks, predictions all-gathered (RCCL over xGMI on GPUs; any torch.distributed backend works).

The reference has no distributed path on the OPNet side; its contract for a
sharded run is therefore "N-GPU result == 1-GPU result on the same clips", which holds because clips
are independent.  Sorting / indexing follows the reference's dataset order
(baselines/datasets.py:74 sorted video names; inference_main.py:210-217 name -> global index).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


def shard_size(n_items: int, world: int) -> int:
    return (n_items + world - 1) // world


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block owned by `rank`: [r*ceil(N/W), min(N, (r+1)*ceil(N/W)))."""
    per = shard_size(n_items, world)
    lo = min(n_items, rank * per)
    return lo, min(n_items, lo + per)


def shard_batches(n_items: int, batch_size: int, world: int, rank: int):
    """Whole-batch sharding for TransformerLstm: its attention spans all clips of a minibatch (S = B*T, SURVEY.md
    section 0), so results depend on batch composition and clips may only be sharded at REFERENCE-BATCH
    granularity: the dataset is cut into the same consecutive batches the reference's DataLoader would form
    and rank r takes batches r, r + world, ...  Returns the list of (lo, hi) clip ranges owned by `rank`."""
    n_batches = (n_items + batch_size - 1) // batch_size
    return [(b * batch_size, min(n_items, (b + 1) * batch_size)) for b in range(rank, n_batches, world)]


def all_gather_predictions(local: torch.Tensor, n_total: int, group: Optional[dist.ProcessGroup] = None,
                           async_op: bool = False):
    """local [n_local, ...] (this rank's shard, n_local may be short or zero on the last ranks) ->
    [n_total, ...] on every rank, in global clip order.  Returns (tensor, work|None); when async_op
    the tensor is valid after work.wait()."""
    world = dist.get_world_size(group)
    per = shard_size(n_total, world)
    pad = local.new_zeros((per,) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    out = local.new_empty((world * per,) + tuple(local.shape[1:]))
    work = dist.all_gather_into_tensor(out, pad, group=group, async_op=async_op)
    return out[:n_total], work


def all_reduce_gradients(params, n_local: int, n_global: int, group: Optional[dist.ProcessGroup] = None,
                         async_op: bool = False):
    """Data-parallel gradient exchange for the training step: ONE all-reduce (sum) over the flat fp32
    gradient buffer (1 421 056 floats = 5.68 MB for OPNet).  Each rank's loss is a mean over its own
    n_local clips, so its gradient is weighted by n_local / n_global before the sum - with equal shards
    that is the usual 1/world, with an uneven last shard it still reproduces the single-process mean
    exactly.  Returns (flat, work|None); call `unflatten_gradients` after work.wait()."""
    grads = [p.grad for p in params if p.grad is not None]
    flat = torch.cat([g.reshape(-1) for g in grads]) * (float(n_local) / float(n_global))
    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return flat, work


def unflatten_gradients(params, flat: torch.Tensor) -> None:
    o = 0
    for p in params:
        if p.grad is None:
            continue
        n = p.grad.numel()
        p.grad.copy_(flat[o:o + n].view_as(p.grad))
        o += n
