"""In-kernel timeline of seqx_forward (csrc/seq_xcd_kernels.hip): s_memtime stamps of CU 0 of every XCD, per wave and phase.
    python tools/seqx_probe.py <model> <B> [T]
stamps: 0 iteration start | 1 cell done (wave 0) | 2 x side done | 3 h[t-1] arrived | 4 products done | 5 after the barrier"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectpermanence_amd import ModelsFactory, _lib  # noqa: E402
from synthdata import opnet as synth  # noqa: E402
from tools.stack_time import CFG  # noqa: E402

name, B = sys.argv[1], int(sys.argv[2])
T = int(sys.argv[3]) if len(sys.argv) > 3 else 300
lib = _lib.load()
boxes, _ = synth.make_batch(0, min(B, 32), T)
x = torch.from_numpy(np.tile(synth.boxes5(boxes), ((B + 31) // 32, 1, 1, 1))[:B]).cuda()
m = ModelsFactory.get_model(name, CFG[name]).eval().cuda()
L = m._runner.L
npair = 8 // L
ng = max(1, -(-((B + 3) // 4) // npair))
nph = T * ng
with torch.no_grad():
    for _ in range(3):
        m(x)
    buf = torch.zeros(8 * 4 * nph * 8, dtype=torch.int64, device="cuda")
    lib.opseq_xcd_set_trace(buf.data_ptr())
    m(x)
    torch.cuda.synchronize()
    lib.opseq_xcd_set_trace(None)
tr = buf.cpu().numpy().reshape(8, 4, nph, 8).astype(np.float64)
lo, hi = nph // 3, 2 * nph // 3                        # steady state
for xcd in range(L):
    print(f"XCD {xcd} (layer {xcd % L}):")
    for w in range(4):
        s = tr[xcd, w, lo:hi]
        period = np.diff(tr[xcd, w, lo:hi + 1, 0]).mean() if hi + 1 <= nph else float("nan")
        seg = [np.mean(s[:, k + 1] - s[:, k]) for k in range(5)]
        print(f"  wave {w}: period {period:7.0f} cycles | start->cell {seg[0]:6.0f} | ->x side {seg[1]:6.0f} | ->h arrived {seg[2]:6.0f} | "
              f"->products {seg[3]:6.0f} | ->barrier {seg[4]:6.0f}")
if L == 2:
    lag = tr[1, 0, lo:hi, 0] - tr[0, 0, lo:hi, 0]
    print(f"layer 1 starts phase p {lag.mean():.0f} cycles after layer 0 starts phase p (s_memtime runs at 100 MHz x ?; same clock on both XCDs assumed)")
