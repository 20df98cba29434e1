"""Look for the rare slow small request after large persistent launches: python tools/stall_probe.py [rounds]  (run it under
rocprofv3 --kernel-trace to see whether a kernel or a gap between kernels is long)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectpermanence_amd import ModelsFactory  # noqa: E402
from synthdata import opnet as synth  # noqa: E402

CFG = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda:0")
model = ModelsFactory.get_model("opnet", CFG)
model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.opnet_synth_params(CFG).items()})
model.eval().to(dev)
boxes = torch.from_numpy(synth.make_batch(0, 32, 300)[0]).to(dev)
x400, x16 = boxes.repeat(13, 1, 1, 1)[:400].contiguous(), boxes[:16].contiguous()
slow = []
with torch.no_grad():
    for it in range(rounds):
        model(x400); model(x400)
        torch.cuda.synchronize()
        for k in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); model(x16); e1.record(); torch.cuda.synchronize()
            if e0.elapsed_time(e1) > 2.0:
                slow.append((it, k, round(e0.elapsed_time(e1), 2)))
print("slow small requests (round, index, ms):", slow, "of", rounds * 5)
model.verify_launches()
