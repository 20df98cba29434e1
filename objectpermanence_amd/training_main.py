"""Training driver - mirror of reference baselines/training_main.py:120-252 for every reasoner of
supported_models.TRAINING_SUPPORTED_MODELS (OPNet, OPNetLstmMlp, BaselineLstm, NonLinearLstm, TransformerLstm and their
*_no_labels variants).  Same config keys (configs/training_config.json), same loop semantics: unshuffled loaders (:155-159),
Adam lr (:150), ReduceLROnPlateau(min, factor, patience) stepped on the TRAIN loss (:151,247), per-epoch evaluation of
train and dev sets with mean IoU and containment-masked mean IoU (:32-117, :240-241), best-dev-IoU checkpoint
`<checkpoints_path>/<model>/<dd-mm-yy>_<iou>.pth` holding the plain state_dict (:19-29, :250-252).  Forward / backward /
Adam / post-process / IoU run in the HIP library.

With torch.distributed initialised (not in the reference):
  * every reference minibatch is split over the ranks in balanced contiguous slices and each rank LOADS ONLY ITS SLICE
    (a `batch_sampler` list of its own index slices, `training_batches`) - the input pipeline is not replicated; the gradients are exchanged by one all-reduce over
    the flat bucket (training.train_step), every rank joining every step even when its slice is empty;
  * transformer_lstm* couples the clips of a minibatch (sequence-first attention), so there a rank takes WHOLE reference
    batches (k*W + r) and one optimiser step spans W of them;
  * the per-epoch evaluation is sharded too and its per-frame IoUs are gathered, so every rank sees the same metrics.
"""
from __future__ import annotations

import time
from datetime import date
from pathlib import Path
from typing import Any, Dict

import numpy as np
import torch
import torch.distributed as dist
from torch.optim.lr_scheduler import ReduceLROnPlateau
from torch.utils import data

from . import metrics, parallel
from .datasets import DatasetsFactory, make_loader
from .launch_monitor import DeferredConsumer, HostEvent as _HostEvent, verify_launches
from .models_factory import ModelsFactory
from .optim import FusedAdam
from .supported_models import DOUBLE_OUTPUT_MODELS
from .training import compute_loss, global_loss, step_aborted, step_skipped_nonfinite, train_step


def save_checkpoint(model: torch.nn.Module, model_name: str, dev_iou: float, checkpoint_dir: str) -> str:
    """training_main.py:19-29"""
    path = Path(checkpoint_dir) / model_name
    path.mkdir(parents=True, exist_ok=True)
    f = path / f"{date.today().strftime('%d-%m-%y')}_{dev_iou}.pth"
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, f)
    print(f"Saved best model so far on dev set with type {model_name} and performance mean IoU of: {dev_iou}")
    return str(f)


def masked_mean_iou(iou: torch.Tensor, cm: torch.Tensor) -> float:
    """Mean over videos of the mean IoU over the video's containment frames.  The reference takes np.mean of a pandas
    Series (training_main.py:105-112) = Series.mean(skipna=True): a video without containment frames is NaN there and
    is SKIPPED, not propagated; NaN only if no video has any."""
    cnt = cm.sum(dim=1)
    masked = torch.where(cm, iou, torch.zeros_like(iou)).sum(dim=1) / cnt.clamp(min=1)
    keep = cnt > 0
    return float(masked[keep].mean()) if bool(keep.any()) else float("nan")


def inference_and_iou_comp(model_name: str, model: torch.nn.Module, device: torch.device, dataset, batch_size: int,
                           num_workers: int):
    """training_main.py:32-117: average loss, dataset mean IoU, containment-masked mean IoU.  Under data parallelism every
    rank evaluates its share of the minibatches (parallel.plan_inference_batches) and the per-frame IoUs / masks / loss
    sums are gathered, so all ranks return the same numbers."""
    world, rank, exchange = parallel.world_rank()
    n_total = len(dataset)
    batches = parallel.plan_inference_batches(model_name, n_total, batch_size, world, rank)
    loader = make_loader(dataset, batches, device, num_workers)
    model.eval()
    loss_sum = torch.zeros((), dtype=torch.float64, device=device)
    ious, contain = [], []

    def consume(output, labels, mask):
        """one minibatch's outputs -> its loss share, per-frame IoUs and containment mask; the outputs themselves are dropped"""
        nonlocal loss_sum
        loss, _, _ = compute_loss(model_name, output, labels, mask, with_consistency=False)
        _, _, iou = metrics.postprocess_and_iou(output, labels)
        ious.append(iou)
        contain.append(torch.sum(mask, dim=-1).type(torch.bool))
        loss_sum += loss.double() * output.shape[0]

    # No sync per minibatch and none that holds the data set: a minibatch is post-processed as soon as its launch is seen
    # complete and clean (an aborted persistent launch is re-run on the launch chain, into the same output tensors, before
    # anything is derived from its output); at most 16 minibatches' outputs are alive at any time
    deferred = DeferredConsumer(model, consume)
    with torch.no_grad():
        for (boxes, _), (labels, mask), _names in loader:
            boxes, labels, mask = boxes.to(device), labels.to(device), mask.to(device)
            out = model(boxes)
            ready = torch.cuda.Event() if device.type == "cuda" else _HostEvent()
            ready.record()
            deferred.add(ready, out[0] if model_name in DOUBLE_OUTPUT_MODELS else out, labels, mask)
        deferred.drain(block=True, all_=True)
        verify_launches(model)
    t_frames = ious[0].shape[1] if ious else 300
    iou = torch.cat(ious) if ious else torch.zeros((0, t_frames), dtype=torch.float64, device=device)
    cm = torch.cat(contain) if contain else torch.zeros((0, t_frames), dtype=torch.bool, device=device)
    if exchange:
        index = torch.tensor([i for b in batches for i in b], dtype=torch.int64, device=device)
        iou = parallel.all_gather_by_index(iou, index, n_total)
        cm = parallel.all_gather_by_index(cm.to(torch.uint8), index, n_total).bool()
        dist.all_reduce(loss_sum)
    return float(loss_sum) / max(n_total, 1), float(iou.mean(dim=1).mean()) if n_total else float("nan"), masked_mean_iou(iou, cm)


def training_batches(model_name: str, n_items: int, batch_size: int, world: int, rank: int):
    """-> (per optimiser step: the dataset indices THIS rank trains on (possibly none), the clips of the whole step).
    Clip-independent reasoners: step k = reference batch k, split in balanced slices.  transformer_lstm*: step k = the W
    reference batches k*W .. k*W + W-1, rank r taking batch k*W + r whole."""
    n_batches = (n_items + batch_size - 1) // batch_size
    span = lambda b: (b * batch_size, min(n_items, (b + 1) * batch_size))
    steps = []
    if parallel.couples_clips(model_name):
        for k in range(0, n_batches, world):
            mine = span(k + rank) if k + rank < n_batches else (0, 0)
            total = sum(span(b)[1] - span(b)[0] for b in range(k, min(k + world, n_batches)))
            steps.append((list(range(*mine)), total))
    else:
        for b in range(n_batches):
            lo, hi = span(b)
            a, z = parallel.balanced_range(hi - lo, world, rank)
            steps.append((list(range(lo + a, lo + z)), hi - lo))
    return steps


@parallel.bounded_host_threads
def training_main(model_name: str, train_config: Dict[str, Any], model_config: Dict[str, int]) -> Dict[str, Any]:
    # the JSON's device (training_main.py:144) - or cuda:LOCAL_RANK when this process is one rank of a torchrun job
    device = parallel.resolve_device(train_config["device"])
    train_ds = DatasetsFactory.get_training_dataset(model_name, train_config["train_sample_dir"], train_config["train_labels_dir"],
                                                    train_config["train_containment_file"])
    dev_ds = DatasetsFactory.get_training_dataset(model_name, train_config["dev_sample_dir"], train_config["dev_labels_dir"],
                                                  train_config["dev_containment_file"])
    bs, nw = train_config["batch_size"], train_config["num_workers"]
    ibs = train_config["inference_batch_size"]
    model = ModelsFactory.get_model(model_name, model_config).to(device)
    parallel.broadcast_parameters(model)          # data parallel: every rank trains rank 0's random initialisation
    optimizer = FusedAdam(model.parameters(), lr=train_config["learning_rate"])
    scheduler = ReduceLROnPlateau(optimizer, mode="min", factor=train_config["lr_scheduler_factor"],
                                  patience=train_config["lr_scheduler_patience"])
    world, rank, exchange = parallel.world_rank()
    steps = training_batches(model_name, len(train_ds), bs, world, rank)
    # this rank's loader yields only its own non-empty slices, in step order
    training_loader = make_loader(train_ds, [idx for idx, _ in steps if idx], device, nw)
    comm = torch.cuda.Stream(device=device) if exchange and device.type == "cuda" else None

    if exchange and parallel.couples_clips(model_name) and world > 1 and rank == 0:
        # (the reference has one process: one optimiser step per minibatch)
        print(f"data parallel over {world} ranks with {model_name}: attention couples the clips of a minibatch, so every rank takes "
              f"WHOLE reference minibatches and one optimiser step consumes {world} of them - effective batch {world * bs} at the "
              f"configured learning rate, {world}x fewer updates per epoch than the reference's single process")
    highest_dev_iou, best_path, history = 0.0, None, []
    start = time.time()
    for epoch in range(train_config["num_epochs"]):
        model.train(mode=True)
        running, consumed = 0.0, 0
        it = iter(training_loader)

        def fetch(step_idx):
            """host-to-device copies of step `step_idx` (None past the end / for an empty slice)"""
            if step_idx >= len(steps) or not steps[step_idx][0]:
                return None
            (boxes, _), (labels, mask), _ = next(it)
            return boxes.to(device, non_blocking=True), labels.to(device, non_blocking=True), mask.to(device, non_blocking=True)

        nxt = fetch(0)
        for k, (_, n_global) in enumerate(steps):
            cur, box = nxt, {}
            # the next step's batch is fetched while the gradient all-reduce of this one is in flight
            args = cur if cur is not None else (None, None, None)
            loss = train_step(model_name, model, optimizer, *args, n_global=n_global,
                              comm_stream=comm, overlap=lambda: box.setdefault("n", fetch(k + 1)))
            nxt = box.get("n")
            loss_value = float(global_loss(model, loss))   # the reference's per-step .item() (training_main.py:212): the sync point
            if step_aborted(model):
                # a persistent launch of this step gave up: the guarded Adam kept the weights; repeat it on the launch chain
                optimizer.rollback_step_count()
                loss_value = float(global_loss(model, train_step(model_name, model, optimizer, *args, n_global=n_global,
                                                                 comm_stream=comm)))
            # a non-finite loss on any rank: the update was skipped on every rank (guarded Adam); counters rolled back, warned
            step_skipped_nonfinite(model, optimizer, loss_value)
            running += loss_value
            consumed += n_global                 # clips this step actually trained on, over all ranks (the reference: (k + 1) * batch_size)
            if (k + 1) % train_config["print_step"] == 0 and rank == 0:
                print("Train Epoch: {} [{}/{}]\t Average Loss: {:.4f} Training began {} seconds ago".format(
                    epoch + 1, consumed, len(train_ds), running / train_config["print_step"], int(time.time() - start)))
                running = 0.0
        train_loss, train_miou, train_cmiou = inference_and_iou_comp(model_name, model, device, train_ds, ibs, nw)
        dev_loss, dev_miou, dev_cmiou = inference_and_iou_comp(model_name, model, device, dev_ds, ibs, nw)
        if rank == 0:
            print("Epoch {} Training Set: Loss {:.4f}, Mean IoU {:.6f}, Mask Mean Iou {:.6f}".format(epoch + 1, train_loss, train_miou, train_cmiou))
            print("Epoch {} Dev Set: Loss {:.4f}, Mean IoU {:.6f}, Mask Mean Iou {:.6f}".format(epoch + 1, dev_loss, dev_miou, dev_cmiou))
        scheduler.step(train_loss)
        history.append({"epoch": epoch + 1, "train_loss": train_loss, "train_miou": train_miou, "dev_loss": dev_loss,
                        "dev_miou": dev_miou, "dev_containment_miou": dev_cmiou, "lr": optimizer.param_groups[0]["lr"]})
        if dev_miou > highest_dev_iou:
            highest_dev_iou = dev_miou
            if rank == 0:
                best_path = save_checkpoint(model, model_name, round(highest_dev_iou, 3), train_config["checkpoints_path"])
    return {"history": history, "best_dev_iou": highest_dev_iou, "checkpoint": best_path}
