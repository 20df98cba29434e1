"""Scratch timing of the OPNet forward on cuda:0 (not the bench contract; see bench.py)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from objectpermanence_amd import ModelsFactory
from oracle import synth

cfg = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}
m = ModelsFactory.get_model("opnet", cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.opnet_synth_params(cfg).items()})
m.eval().to("cuda:0")
for B in [int(a) for a in sys.argv[1:]] or [32]:
    boxes = torch.from_numpy(np.tile(synth.make_batch(0, 4, 300)[0], ((B + 3) // 4, 1, 1, 1))[:B]).cuda()
    for graph in (True, False):
        m.use_graph = graph
        with torch.no_grad():
            for _ in range(3):
                m(boxes)
            torch.cuda.synchronize()
            n = 10
            t0 = time.perf_counter()
            for _ in range(n):
                m(boxes)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
        print(f"B={B} graph={graph}: {dt*1e3:.3f} ms/forward  {B/dt:.0f} clips/s  {dt/303*1e6:.2f} us/step", flush=True)
