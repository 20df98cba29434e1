"""Video-level data parallelism for the reasoners: one process per GPU, clips sharded in contiguous
blocks, predictions all-gathered (RCCL over xGMI on GPUs; any torch.distributed backend works).

The reference has no distributed path on the OPNet side; its contract for a
sharded run is therefore "N-GPU result == 1-GPU result on the same clips", which holds because clips
are independent.  Sorting / indexing follows the reference's dataset order
(baselines/datasets.py:74 sorted video names; inference_main.py:210-217 name -> global index).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import torch
import torch.distributed as dist

FORCE_ENV = "OPNET_FORCE_DIST"      # "1": take every data-parallel branch even in a process group of ONE rank


def forced() -> bool:
    return os.environ.get(FORCE_ENV, "0") == "1"


def is_active(group: Optional[dist.ProcessGroup] = None) -> bool:
    """Is the data-parallel exchange to be run?  Yes in an initialised process group of more than one rank - and in a group
    of one rank when OPNET_FORCE_DIST=1: the whole device side of the exchange (comm stream, in-place gradient bucket, RCCL
    collectives, guard slot, gathers by index) then executes on a single GPU, which is how the `-m gpu` tests cover it."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size(group) > 1 or forced()


def world_rank(group: Optional[dist.ProcessGroup] = None) -> Tuple[int, int, bool]:
    """(world size, rank, exchange active?) of this process: (1, 0, False) outside torch.distributed"""
    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0, False
    return dist.get_world_size(group), dist.get_rank(group), is_active(group)


class Launch:
    """What `init_from_env` found: did this process join a group it must leave again, and which device is its own."""

    def __init__(self, owned: bool, device: Optional[torch.device], world: int, rank: int, local_rank: int):
        self.owned, self.device, self.world, self.rank, self.local_rank = owned, device, world, rank, local_rank

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        # An exception is propagating (a refused clip file, an out-of-memory, an ABI error): this rank must DIE, not meet the
        # others at a barrier they will never reach - they are inside a gradient all-reduce or a gather of another size, the
        # mismatched collectives would hang until the watchdog fires, and torchrun only reaps the peers once this process has
        # exited with its traceback.  The barrier belongs to the clean path only.
        shutdown(self, failed=exc[0] is not None)
        return False


_launch: Optional[Launch] = None


def init_from_env(backend: Optional[str] = None) -> Launch:
    """The product entry points' way into data parallelism (`python -m torch.distributed.run --nproc-per-node N -m
    objectpermanence_amd training ...`; the reference has one device per run from its JSON, training_main.py:144,
    inference_main.py:189-207).  With WORLD_SIZE in the environment (torchrun's contract: RANK, LOCAL_RANK, WORLD_SIZE,
    MASTER_ADDR, MASTER_PORT) - or OPNET_FORCE_DIST=1 for a group of one - the process joins the job: backend "nccl"
    (= RCCL) bound to `cuda:LOCAL_RANK` when a GPU is visible, "gloo" otherwise (host-logic tests), and the device every
    driver then uses is THAT one, whatever the JSON config says (`resolve_device`).  Without either it does nothing.
    A group somebody else initialised is used as it is and not destroyed."""
    global _launch
    world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    rank = int(os.environ.get("RANK", "0") or 0)
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)) or 0)
    has_gpu = torch.cuda.is_available()
    device = torch.device("cuda", local_rank) if has_gpu else None
    if dist.is_available() and dist.is_initialized():
        _launch = Launch(False, device if world else None, dist.get_world_size(), dist.get_rank(), local_rank)
        return _launch
    if world <= 0 and not forced():
        _launch = Launch(False, None, 1, 0, 0)
        return _launch
    world = max(world, 1)
    if has_gpu:
        if local_rank >= torch.cuda.device_count():      # (with gloo, LOCAL_RANK may repeat: ranks sharing a device)
            raise RuntimeError(f"LOCAL_RANK {local_rank} but this node exposes {torch.cuda.device_count()} GPU(s): start at most "
                               "one rank per GPU")
        torch.cuda.set_device(device)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        if world > 1:
            # launchers that export only RANK / WORLD_SIZE / MASTER_ADDR (srun, mpirun wrappers): one deterministic default that every
            # rank derives alike (README: export MASTER_PORT to run two such jobs on one node)
            os.environ["MASTER_PORT"] = "29541"
        else:
            # a forced group of one: any free port (a fixed one would collide between two such runs on one box)
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    # OPNET_DIST_BACKEND=gloo: several ranks on ONE device (RCCL refuses that) - how a box with a single GPU runs world size 2
    # (tests/test_dp_two_ranks_gpu.py); gloo moves device tensors through the host
    backend = backend or os.environ.get("OPNET_DIST_BACKEND") or ("nccl" if has_gpu else "gloo")
    kw = {"device_id": device} if backend == "nccl" else {}
    dist.init_process_group(backend=backend, rank=rank, world_size=world, **kw)
    _launch = Launch(True, device, world, rank, local_rank)
    return _launch


def shutdown(launch: Optional[Launch] = None, failed: bool = False) -> None:
    """leave the group `init_from_env` joined.  Clean path: all ranks meet first, so that no rank tears the communicator down
    under a collective another rank is still in.  failed: this rank is on its way out with an exception - no barrier, no
    orderly destroy (both would wait for peers that are in other collectives): the communicator is aborted where the backend
    offers it and the process goes on to exit non-zero, which is what lets the launcher kill the other ranks."""
    global _launch
    launch = launch or _launch
    if launch is not None and launch.owned and dist.is_initialized():
        if failed:
            try:                             # (nccl: tears the communicator down without waiting for its peers; a no-op elsewhere)
                from torch.distributed.distributed_c10d import _abort_process_group
            except ImportError:
                _abort_process_group = None
            aborted = False
            if _abort_process_group is not None:
                try:
                    _abort_process_group()
                    aborted = True
                except Exception:
                    pass
            if not aborted and dist.get_backend() == "nccl":
                # a torch build without the (private) abort: the communicator cannot be torn down without its peers, and its teardown at
                # interpreter exit would block on them - leave at once; the exception that brought us here has been printed by the caller
                import sys
                import traceback
                traceback.print_exc()
                sys.stderr.flush()
                os._exit(1)
        else:
            try:
                dist.barrier()
            finally:
                dist.destroy_process_group()
        launch.owned = False
    if launch is _launch:
        _launch = None


def resolve_device(config_device) -> torch.device:
    """The device a driver runs on: the JSON's `device` (training_main.py:144, inference_main.py:189) - unless this process
    was started as one rank of a job (`init_from_env`), where it is `cuda:LOCAL_RANK`: eight ranks that all read
    "cuda:0" from the same config file would otherwise share one GPU."""
    if _launch is not None and _launch.device is not None and dist.is_available() and dist.is_initialized():
        return _launch.device
    return torch.device(config_device)


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group: Optional[dist.ProcessGroup] = None) -> None:
    """every rank starts from rank `src`'s weights: the reference builds its model with torch's random initialisation
    (training_main.py:144-150, one process) - N processes doing the same would hold N different models, and averaging their
    gradients would train none of them.  In place, before the first forward (nothing is packed yet).  No-op outside a job."""
    if not is_active(group):
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t.data, src, group=group)


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


def balanced_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """np.array_split-style block of `rank`: sizes differ by at most one and no rank is empty while n_items >= world
    (the per-batch split of a TRAINING minibatch: with ceil(N/W) blocks a last batch of 5 clips on 4 ranks would leave
    rank 3 without work)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def couples_clips(model_name: str) -> bool:
    """TransformerLstm attends over all clips of a minibatch (S = B*T, reference learned_models.py:183-185; SURVEY.md
    section 0): its outputs depend on batch composition, so it may only be sharded in whole reference batches."""
    return model_name.startswith("transformer_lstm")


def plan_inference_batches(model_name: str, n_items: int, batch_size: int, world: int, rank: int):
    """The minibatches (lists of dataset indices) `rank` evaluates.  Clip-independent reasoners: the contiguous block
    shard_range() cut into batches (any cut gives the same outputs).  TransformerLstm: exactly the batches the reference's
    DataLoader would form over the WHOLE dataset, dealt round-robin (shard_batches) - never a split batch."""
    if couples_clips(model_name):
        return [list(range(lo, hi)) for lo, hi in shard_batches(n_items, batch_size, world, rank)]
    lo, hi = shard_range(n_items, world, rank)
    return [list(range(b, min(hi, b + batch_size))) for b in range(lo, hi, batch_size)]


def all_gather_by_index(local: torch.Tensor, index: torch.Tensor, n_total: int,
                        group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """local [n_local, ...] with the global dataset index of every row -> [n_total, ...] in dataset order on every rank
    (ranks may own any subset; n_local may be zero).  Two collectives: the padded rows and their indices."""
    world = dist.get_world_size(group)
    counts = [torch.zeros(1, dtype=torch.int64, device=local.device) for _ in range(world)]
    dist.all_gather(counts, torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device), group=group)
    per = max(int(max(c.item() for c in counts)), 1)
    pad = local.new_zeros((per,) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    ipad = torch.full((per,), -1, dtype=torch.int64, device=local.device)
    ipad[: local.shape[0]] = index.to(local.device)
    rows = local.new_empty((world * per,) + tuple(local.shape[1:]))
    idx = torch.empty(world * per, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(rows, pad, group=group)
    dist.all_gather_into_tensor(idx, ipad, group=group)
    keep = idx >= 0
    out = local.new_zeros((n_total,) + tuple(local.shape[1:]))
    out[idx[keep]] = rows[keep]
    return out


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


class GradBucket:
    """ONE flat fp32 buffer behind the gradients of a FIXED parameter list (1 421 056 floats = 5.68 MB for OPNet): the
    backward pass writes every weight gradient straight into its slice (learned_models._OPNetTrainFunction when the module
    carries the bucket), the data-parallel exchange is one all-reduce over `flat`, and the optimiser reads the reduced
    slices - no torch.cat before and no copy back after the collective.  The list is fixed at construction, so every rank
    reduces the same number of elements whatever gradients it happened to produce (a rank with an empty shard contributes
    zeros and still joins)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        ref = self.params[0]
        # 4 extra floats behind the gradients, the GUARD: [n] = this rank's "my persistent launches gave up" flag, [n+1] =
        # "my loss is not finite" (opnet_dp_guard_f32 writes both), summed over the ranks by the same all-reduce, so that every
        # rank's optimiser skips the step when any rank's gradients are unusable (optim.FusedAdam.guard_ptr); the rest is
        # padding (16-byte multiple)
        self.n_grad = n
        self._buf = torch.zeros(n + 4, dtype=torch.float32, device=ref.device)
        self.flat = self._buf[:n]
        self.guard = self._buf[n:n + 4]
        self.offsets, o = [], 0
        for p in self.params:
            self.offsets.append(o)
            o += p.numel()

    def view(self, i: int) -> torch.Tensor:
        p, o = self.params[i], self.offsets[i]
        return self.flat[o:o + p.numel()].view_as(p)       # a fresh tensor object on the bucket's storage

    def collect(self, fill_missing: bool = True) -> None:
        """make `flat` hold the current gradients: slices the backward already wrote in place are left alone, anything
        else (a sibling model's autograd-made gradient) is copied; then p.grad aliases it.  A parameter WITHOUT a gradient:
        fill_missing (data parallel - every rank must reduce the same elements and take the same optimiser steps) gives it
        zeros; otherwise it keeps grad = None and the optimiser skips it, as torch.optim.Adam does (training_main.py:217)."""
        for i, p in enumerate(self.params):
            v = self.view(i)
            if p.grad is None:
                if not fill_missing:
                    continue
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                v.copy_(p.grad)
            p.grad = v

    def all_reduce(self, n_local: int, n_global: int, group: Optional[dist.ProcessGroup] = None, async_op: bool = False):
        """weighted n_local / n_global so that an uneven split still reproduces the single-process mean-loss gradient
        (SURVEY.md 8-e1); call on the stream the collective should run on"""
        self.flat.mul_(float(n_local) / float(n_global))
        return dist.all_reduce(self._buf, op=dist.ReduceOp.SUM, group=group, async_op=async_op)     # gradients + guard


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


def bounded_host_threads(fn):
    """Decorator of the driver entry points (reasoning_inference_main, cater_setup_inference, training_main).  Their host-side tensor
    work is per-minibatch bookkeeping on a few KB (index vectors, masks, int32 boxes, the IoU means); torch's default intra-op pool -
    one thread per core, 128 on the MI355X host - turns every such op into an OpenMP fork / join over all of them, which also fights
    the clip-file reader's threads for the cores.  Measured on the box (tools/e2e_inference_time.py, 4 096 clips from files, 12
    reader threads): 5.4 k clips/s with the default pool, 35.7 k with 8 threads, 34.8 k with 1.  The pool is bounded for the call
    (OPNET_HOST_THREADS, default 8; 0 = leave it alone) and restored afterwards."""
    import functools

    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        want = int(os.environ.get("OPNET_HOST_THREADS", "8"))
        prev = torch.get_num_threads()
        if want > 0 and prev > want:
            torch.set_num_threads(want)
        try:
            return fn(*args, **kwargs)
        finally:
            if torch.get_num_threads() != prev:
                torch.set_num_threads(prev)
    return wrapped
