"""The reference's one-frame detector call (baselines/detector.py:71-86): how much of its wall time is host work (Python + ctypes +
allocations per launch) and how much GPU time.  python tools/detector_host_time.py [--profile]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from objectpermanence_amd.detector import CaterObjectDetector  # noqa: E402
from oracle import detector_oracle as do  # noqa: E402

det = CaterObjectDetector(None)
det.load_state_dict({**do.synth_backbone_params(), **do.synth_head_params()}, "cuda:0")
dev = torch.device("cuda:0")
f = np.random.default_rng(0).integers(0, 256, size=(240, 320, 3), dtype=np.uint8)
for _ in range(3):
    det(f, dev)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
ts = []
for _ in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0.record()
    out = det(f, dev)
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3, e0.elapsed_time(e1)))
med = [round(float(np.median([t[k] for t in ts])), 3) for k in range(3)]
print(f"one-frame call: host returns after {med[0]} ms, results ready after {med[1]} ms, first-to-last kernel {med[2]} ms (median of 20)")
if "--profile" in sys.argv:
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        det(f, dev)
    torch.cuda.synchronize()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(20)
