import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from objectpermanence_amd.detector import ResNet50FPNBackbone, preprocess_frame
from oracle import detector_oracle as do
bb = ResNet50FPNBackbone(do.synth_backbone_params(), "cuda:0")
frame = np.random.default_rng(0).integers(0, 256, size=(240, 320, 3), dtype=np.uint8)
for _ in range(2):
    f = bb.forward_nhwc(preprocess_frame(frame))
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    f = bb.forward_nhwc(preprocess_frame(frame))
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
# MACs of ResNet-50 + FPN at 800x1088: count from layer shapes
print(f"240x320 frame -> 800x1088: preprocess + ResNet-50-FPN backbone {dt*1e3:.2f} ms/frame ({1/dt:.1f} frames/s); maps:", {k: tuple(v.shape) for k, v in f.items()})
