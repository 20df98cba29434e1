"""Randomised shape sweep of the per-XCD persistent forward (opnet_xcd_forward, both head forms, ring and full-history layouts)
against the numpy oracle - not part of the test suite; run on the GPU box:  CASES=40 SEED=0 python tools/fuzz_xcd.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from objectpermanence_amd import ModelsFactory
from oracle import opnet_oracle as oo, synth

CFG = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}
rng = np.random.default_rng(int(os.environ.get("SEED", "0")))
worst = {"y": 0.0, "logits": 0.0}
for case in range(int(os.environ.get("CASES", "40"))):
    B = int(rng.choice([33, 48, 64, 65, 100, 128, 129, 200, 256, 320, 383, 384, 400, 512, 640, 777, 1024]))
    T = int(rng.choice([1, 2, 3, 4, 5, 6, 7, 9, 17, 40]))
    ho = str(int(rng.integers(0, 2)))
    ring = str(int(rng.integers(0, 2)))
    os.environ["OPNET_XCD_HO"], os.environ["OPNET_XCD_RING"] = ho, ring
    p = synth.opnet_synth_params(CFG, salt=case)
    boxes, _ = synth.make_batch(2000 + case, min(B, 96), T)
    boxes = np.tile(boxes, ((B + boxes.shape[0] - 1) // boxes.shape[0], 1, 1, 1))[:B].copy()
    boxes[:, :, :, :4] += (np.arange(B, dtype=np.float32) % 7)[:, None, None, None] * 1e-3      # no two clips alike
    m = ModelsFactory.get_model("opnet", CFG)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in p.items()})
    m.eval().to("cuda:0")
    m.use_xcd = "1"
    with torch.no_grad():
        y, lg = m(torch.from_numpy(boxes).cuda())
    torch.cuda.synchronize()
    assert m.verify_launches() == 0, "a persistent launch aborted"
    ry, rl = oo.opnet_forward(boxes, p, np.float64)
    ey, el = float(np.abs(y.cpu().numpy() - ry).max()), float(np.abs(lg.cpu().numpy() - rl).max())
    worst["y"], worst["logits"] = max(worst["y"], ey), max(worst["logits"], el)
    ok = ey < 2e-5 and el < 1e-4
    print(f"case {case}: B={B} T={T} head_once={ho} ring={ring} |dy| {ey:.2e} |dlogits| {el:.2e} {'ok' if ok else 'FAIL'}", flush=True)
    assert ok
print("WORST", {k: f"{v:.3e}" for k, v in worst.items()})
