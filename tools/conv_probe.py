"""Time single conv shapes through opdet_conv2d_f32: python tools/conv_probe.py [name ...]; shapes as in the detector at
16 frames per pass.  Prints TFLOP/s from event timing; run under rocprofv3 --pmc for MFMA busy cycles."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from objectpermanence_amd.detector import _Conv

SHAPES = {  # name: (N, H, W, Cin, Cout, k, stride, pad)
    "outer0": (16, 200, 272, 256, 256, 3, 1, 1),
    "outer2": (16, 50, 68, 256, 256, 3, 1, 1),
    "l1c3": (16, 200, 272, 64, 256, 1, 1, 0),
    "l2c2": (16, 100, 136, 128, 128, 3, 1, 1),
    "l3c1": (16, 50, 68, 1024, 256, 1, 1, 0),
    "l4c2": (16, 25, 34, 512, 512, 3, 1, 1),
    "fc6": (1, 1, 16000, 12544, 1024, 1, 1, 0),
    "inner0": (16, 200, 272, 256, 256, 1, 1, 0),
}
reps = int(os.environ.get("REPS", "10"))
for name in sys.argv[1:] or list(SHAPES):
    n, h, w, cin, cout, k, s, p = SHAPES[name]
    wt = torch.randn(cout, cin, k, k) * (1.0 / (cin * k * k) ** 0.5)
    conv = _Conv({"w": wt, "b": torch.zeros(cout)}, "w", bias="b", stride=s, pad=p)
    x = torch.randn(n, h, w, cin, device="cuda:0")
    for _ in range(2):
        y = conv(x, relu=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        y = conv(x, relu=True)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    fl = 2.0 * y.numel() * k * k * cin
    print(f"{name:8s} M={y.numel() // cout:8d} Cout={cout:5d} K={k * k * cin:6d}  {ms * 1e3:9.1f} us  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)
