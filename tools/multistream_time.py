"""Scratch: N independent B=32 forwards in flight on N HIP streams (own workspace + graph each)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from objectpermanence_amd import ModelsFactory
from oracle import synth
cfg = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}
params = synth.opnet_synth_params(cfg)
B = 32
boxes = torch.from_numpy(synth.make_batch(0, B, 300)[0]).cuda()
for ns in (1, 2, 3, 4, 6, 8):
    models = []
    for i in range(ns):
        m = ModelsFactory.get_model("opnet", cfg)
        m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
        models.append(m.eval().to("cuda:0"))
    streams = [torch.cuda.Stream() for _ in range(ns)]
    def run(n):
        outs = []
        for it in range(n):
            i = it % ns
            with torch.cuda.stream(streams[i]), torch.no_grad():
                outs.append(models[i](boxes))
        return outs
    run(2 * ns); torch.cuda.synchronize()
    n = 40
    t0 = time.perf_counter(); outs = run(n); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ok = all(torch.equal(o[0], outs[0][0]) for o in outs)
    print(f"streams={ns}: {n*B/dt:.0f} clips/s  ({dt/n*1e3:.3f} ms per forward amortised) identical={ok}", flush=True)
