"""Full detector (backbone + RPN + RoI heads + post-process) timing on 240x320 frames; stage breakdown with events."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from objectpermanence_amd.detector import CaterObjectDetector, preprocess_frame, resized_size
from oracle import detector_oracle as do

det = CaterObjectDetector(None)
det.load_state_dict({**do.synth_backbone_params(), **do.synth_head_params()}, "cuda:0")
dev = torch.device("cuda:0")
frames = [f for f in np.random.default_rng(0).integers(0, 256, size=(16, 240, 320, 3), dtype=np.uint8)]
for nb in [int(a) for a in sys.argv[1:]] or [1, 16]:
    fs = frames[:nb]
    run = (lambda: det(fs[0], dev)) if nb == 1 else (lambda: det.detect_batch(fs, dev))
    for _ in range(2):
        out = run()
    torch.cuda.synchronize()
    n = max(2, 16 // nb)
    t0 = time.perf_counter()
    for _ in range(n):
        out = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print(f"{nb} frame(s) per call: {dt*1e3:.2f} ms = {dt/nb*1e3:.2f} ms/frame ({nb/dt:.1f} frames/s); detections/frame:",
          [len(o["scores"]) for o in out][:4], flush=True)

# stage breakdown for one frame
x = preprocess_frame(frames[0], dev)
image_size = resized_size(240, 320)
ev = lambda: torch.cuda.Event(enable_timing=True)
names, marks = [], []
def mark(name):
    e = ev(); e.record(); names.append(name); marks.append(e)
for rep in range(3):
    names, marks = [], []
    mark("start")
    feats = det.backbone.forward_nhwc(x); mark("backbone")
    head = det.heads.rpn_head(feats); mark("rpn_head")
    props, ps, count = det.heads.proposals(head, image_size, x.shape[1:3]); mark("proposals")
    pooled = det.heads.roi_align(list(feats.values()), props, count, image_size); mark("roi_align")
    cls, reg = det.heads.box_heads(pooled); mark("box_heads")
    o = det.heads.detections(cls, reg, props, count, image_size, (240, 320)); mark("detections")
    torch.cuda.synchronize()
print("stages (ms):", {n: round(marks[i - 1].elapsed_time(marks[i]), 3) for i, n in enumerate(names) if i}, "proposals:", int(count.item()))
