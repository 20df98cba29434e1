"""Training driver - mirror of reference baselines/training_main.py:120-252 for the reasoners whose training
path is built (OPNet / opnet_no_labels).  Same config keys (configs/training_config.json), same loop
semantics: unshuffled loaders (:155-159), Adam lr (:150), ReduceLROnPlateau(min, factor, patience) stepped
on the TRAIN loss (:151,247), per-epoch evaluation of train and dev sets with mean IoU and containment-masked
mean IoU (:32-117, :240-241), best-dev-IoU checkpoint `<checkpoints_path>/<model>/<dd-mm-yy>_<iou>.pth` holding
the plain state_dict (:19-29, :250-252).  Forward/backward/Adam/post-process/IoU run in the HIP library; with
torch.distributed initialised the training batches are split over the ranks (training.train_step)."""
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
from .datasets import DatasetsFactory
from .models_factory import ModelsFactory
from .optim import FusedAdam
from .supported_models import DOUBLE_OUTPUT_MODELS
from .training import compute_loss, train_step


def save_checkpoint(model: torch.nn.Module, model_name: str, dev_iou: float, checkpoint_dir: str) -> str:
    """training_main.py:19-29"""
    path = Path(checkpoint_dir) / model_name
    path.mkdir(parents=True, exist_ok=True)
    f = path / f"{date.today().strftime('%d-%m-%y')}_{dev_iou}.pth"
    torch.save({k: v.detach().cpu() for k, v in model.state_dict().items()}, f)
    print(f"Saved best model so far on dev set with type {model_name} and performance mean IoU of: {dev_iou}")
    return str(f)


def inference_and_iou_comp(model_name: str, model: torch.nn.Module, device: torch.device, loader: data.DataLoader):
    """training_main.py:32-117: average loss, dataset mean IoU, containment-masked mean IoU (a video whose mask is
    empty contributes NaN to the containment mean, exactly like np.mean over the reference's DataFrame column)."""
    model.eval()
    total_loss, n_seen = 0.0, 0
    ious, contain = [], []
    with torch.no_grad():
        for (boxes, _), (labels, mask), _names in loader:
            boxes, labels, mask = boxes.to(device), labels.to(device), mask.to(device)
            out = model(boxes)
            output = out[0] if model_name in DOUBLE_OUTPUT_MODELS else out
            loss, _, _ = compute_loss(model_name, output, labels, mask)
            _, _, iou = metrics.postprocess_and_iou(output, labels)
            ious.append(iou)
            contain.append(torch.sum(mask, dim=-1).type(torch.bool))
            total_loss += float(loss) * boxes.shape[0]
            n_seen += boxes.shape[0]
    iou = torch.cat(ious)
    cm = torch.cat(contain)
    video_mean = iou.mean(dim=1)
    cnt = cm.sum(dim=1)
    masked = torch.where(cm, iou, torch.zeros_like(iou)).sum(dim=1) / cnt.clamp(min=1)
    masked = torch.where(cnt > 0, masked, torch.full_like(masked, float("nan")))
    return total_loss / max(n_seen, 1), float(video_mean.mean()), float(masked.mean())


def training_main(model_name: str, train_config: Dict[str, Any], model_config: Dict[str, int]) -> Dict[str, Any]:
    device = torch.device(train_config["device"])
    train_ds = DatasetsFactory.get_training_dataset(model_name, train_config["train_sample_dir"], train_config["train_labels_dir"],
                                                    train_config["train_containment_file"])
    dev_ds = DatasetsFactory.get_training_dataset(model_name, train_config["dev_sample_dir"], train_config["dev_labels_dir"],
                                                  train_config["dev_containment_file"])
    bs, nw = train_config["batch_size"], train_config["num_workers"]
    model = ModelsFactory.get_model(model_name, model_config).to(device)
    optimizer = FusedAdam(model.parameters(), lr=train_config["learning_rate"])
    scheduler = ReduceLROnPlateau(optimizer, mode="min", factor=train_config["lr_scheduler_factor"],
                                  patience=train_config["lr_scheduler_patience"])
    training_loader = data.DataLoader(train_ds, batch_size=bs, num_workers=nw)
    infer_cfg = {"batch_size": train_config["inference_batch_size"], "num_workers": nw}
    train_inference_loader = data.DataLoader(train_ds, **infer_cfg)
    dev_loader = data.DataLoader(dev_ds, **infer_cfg)
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank() if world > 1 else 0

    highest_dev_iou, best_path, history = 0.0, None, []
    start = time.time()
    for epoch in range(train_config["num_epochs"]):
        model.train(mode=True)
        running = 0.0
        for batch_idx, ((boxes, _), (labels, mask), _) in enumerate(training_loader, 1):
            n_global = int(boxes.shape[0])
            lo, hi = parallel.shard_range(n_global, world, rank)          # DP: contiguous slice of every batch
            if hi > lo:
                loss = train_step(model_name, model, optimizer, boxes[lo:hi].to(device), labels[lo:hi].to(device),
                                  mask[lo:hi].to(device), n_global=n_global)
                running += float(loss)
            if batch_idx % train_config["print_step"] == 0:
                print("Train Epoch: {} [{}/{}]\t Average Loss: {:.4f} Training began {} seconds ago".format(
                    epoch + 1, batch_idx * bs, len(train_ds), running / train_config["print_step"], int(time.time() - start)))
                running = 0.0
        train_loss, train_miou, train_cmiou = inference_and_iou_comp(model_name, model, device, train_inference_loader)
        dev_loss, dev_miou, dev_cmiou = inference_and_iou_comp(model_name, model, device, dev_loader)
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
