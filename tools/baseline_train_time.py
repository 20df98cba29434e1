"""baseline_lstm / non_linear_lstm training step (fwd + L1 + bwd + Adam) timing: python tools/baseline_train_time.py [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from objectpermanence_amd import ModelsFactory, FusedAdam
from objectpermanence_amd.training import train_step
from oracle import synth
from tools.stack_time import CFG
for name in ("baseline_lstm", "non_linear_lstm"):
    for B in [int(a) for a in sys.argv[1:]] or [32]:
        m = ModelsFactory.get_model(name, CFG[name]).to("cuda:0").train(True)
        opt = FusedAdam(m.parameters(), lr=1e-4)
        b, l = synth.make_batch(0, min(B, 8), 300)
        x = torch.from_numpy(np.tile(synth.boxes5(b), ((B + 7) // 8, 1, 1, 1))[:B].copy()).cuda()
        y = torch.from_numpy(np.tile(l, ((B + 7) // 8, 1, 1))[:B].copy()).cuda()
        for _ in range(3):
            loss = train_step(name, m, opt, x, y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            loss = train_step(name, m, opt, x, y)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print(f"{name} train B={B}: {dt*1e3:.3f} ms/step  {B/dt:.0f} clips/s  loss {float(loss):.4f}", flush=True)
