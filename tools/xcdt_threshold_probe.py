"""Where should the router hand a stacked-LSTM forward to the throughput form?  One eval forward of baseline_lstm (one layer) and of
non_linear_lstm (two layers) at B clips on the 4-clip form (OPSEQ_XCDT_MIN_BATCH above B) and on the throughput form (at or below B).
    python tools/xcdt_threshold_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from objectpermanence_amd import ModelsFactory
from synthdata import opnet as synth
from tools.stack_time import CFG

dev = torch.device("cuda:0")
for name, batches in (("baseline_lstm", (48, 64, 65, 72, 80, 96, 97, 112, 128)), ("non_linear_lstm", (16, 17, 24, 32, 33, 40, 48, 49, 64))):
    for B in batches:
        boxes, _ = synth.make_batch(0, min(B, 32), 300)
        x = torch.from_numpy(np.tile(synth.boxes5(boxes), ((B + 31) // 32, 1, 1, 1))[:B]).to(dev)
        row = []
        for minb in (100000, 1):
            m = ModelsFactory.get_model(name, CFG[name]).eval().to(dev)
            m._runner.XCDT_MIN_BATCH = minb
            with torch.no_grad():
                for _ in range(3):
                    m(x)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    m(x)
                e1.record()
                torch.cuda.synchronize()
            row.append(e0.elapsed_time(e1) / 10)
        print(f"{name} B={B}: 4-clip form {row[0]:.3f} ms   throughput form {row[1]:.3f} ms   -> {'throughput' if row[1] < row[0] else '4-clip'}", flush=True)
