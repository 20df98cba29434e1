import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from objectpermanence_amd import ModelsFactory, metrics
from oracle import synth
cfg = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}
m = ModelsFactory.get_model("opnet", cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.opnet_synth_params(cfg).items()})
m.eval().to("cuda:0")
boxes = torch.from_numpy(synth.make_batch(0, 32, 300)[0]).cuda()
streams = [torch.cuda.Stream() for _ in range(8)]
with torch.no_grad():
    for s in streams:
        with torch.cuda.stream(s): m(boxes)
    torch.cuda.synchronize()
    for what in ("model", "model+post"):
        for graph in (True, False):
            m.use_graph = graph
            for s in streams:
                with torch.cuda.stream(s): m(boxes)
            torch.cuda.synchronize()
            # host enqueue cost: 8 forwards on 8 different streams (no stream is ever backed up)
            t0 = time.perf_counter()
            for s in streams:
                with torch.cuda.stream(s):
                    y, _ = m(boxes)
                    if what != "model": metrics.postprocess_and_iou(y)
            t1 = time.perf_counter()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            print(f"{what} graph={graph}: host enqueue {1e3*(t1-t0)/8:.3f} ms/forward; wall incl. drain {1e3*(t2-t0)/8:.3f} ms/forward")
