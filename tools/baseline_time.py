import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from objectpermanence_amd import ModelsFactory
from oracle import synth
cfg = {"videos_hidden_dim": 512}
m = ModelsFactory.get_model("baseline_lstm", cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.baseline_lstm_synth_params(cfg).items()})
m.eval().to("cuda:0")
B = int(sys.argv[1])
x = torch.from_numpy(np.tile(synth.boxes5(synth.make_batch(0, 4, 300)[0]), ((B + 3) // 4, 1, 1, 1))[:B].copy()).cuda()
with torch.no_grad():
    for _ in range(3): m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m(x)
    torch.cuda.synchronize(); print("ms", (time.perf_counter() - t0) / 10 * 1e3)
