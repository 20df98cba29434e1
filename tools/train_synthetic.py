"""Train OPNet with the HIP training path on synthetic CATER-shaped clips (seeds 1000+c) and save the weights
rounded to fp16 (exactly representable in fp32) -> gpurun_out/opnet_trained_fp16.npz.
The result is committed as tests/golden/opnet_trained_fp16.npz and used for non-vacuous mean-IoU parity tests."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from objectpermanence_amd import ModelsFactory, FusedAdam, metrics
from objectpermanence_amd.training import train_step
from oracle import synth

cfg = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}
torch.manual_seed(0)
model = ModelsFactory.get_model("opnet", cfg).to("cuda:0")
opt = FusedAdam(model.parameters(), lr=1e-3)
sched = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", factor=0.8, patience=2)   # training_main.py:151
n_train, B = int(os.environ.get("N_TRAIN", 512)), 32
epochs = int(os.environ.get("EPOCHS", 30))
boxes_np, labels_np = synth.make_batch(0, n_train, 300)
dev_b, dev_l = synth.make_batch(100000, 64, 300)
boxes, labels = torch.from_numpy(boxes_np).cuda(), torch.from_numpy(labels_np).cuda()
dev_b, dev_l = torch.from_numpy(dev_b).cuda(), torch.from_numpy(dev_l).cuda()

def evaluate():
    model.eval()
    with torch.no_grad():
        y, _ = model(dev_b)
    _, _, iou = metrics.postprocess_and_iou(y, dev_l)
    model.train(True)
    return metrics.mean_iou_and_map(iou)

t0 = time.time()
for ep in range(epochs):
    model.train(True)
    tot = 0.0
    for i in range(0, n_train, B):               # the reference does not shuffle (training_main.py:155-159)
        tot += float(train_step("opnet", model, opt, boxes[i:i + B], labels[i:i + B]))
    tot /= (n_train // B)
    sched.step(tot)
    miou, map50 = evaluate()
    print(f"epoch {ep + 1}: train L1 {tot:.4f}  dev mean-IoU {miou:.4f}  mAP@0.5 {map50:.4f}  lr {opt.param_groups[0]['lr']:.2e}  [{time.time() - t0:.0f}s]", flush=True)
os.makedirs("gpurun_out", exist_ok=True)
sd = {k: v.detach().cpu().numpy().astype(np.float16) for k, v in model.state_dict().items()}
np.savez_compressed("gpurun_out/opnet_trained_fp16.npz", **sd)
print("saved", sum(v.size for v in sd.values()), "weights")
