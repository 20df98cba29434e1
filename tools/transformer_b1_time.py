import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from objectpermanence_amd import ModelsFactory
from oracle import synth
tcfg = {"boxes_features_dim": 256, "num_attention_heads": 4, "num_attention_layers": 2, "num_lstm_layers": 2, "lstm_hidden_dim": 512}
m = ModelsFactory.get_model("transformer_lstm", tcfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.transformer_lstm_synth_params(tcfg).items()})
m.eval().to("cuda:0")
x = torch.from_numpy(synth.make_batch(0, 1, 300)[0][..., :5].copy()).cuda()
with torch.no_grad():
    for _ in range(3): m(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): m(x)
    torch.cuda.synchronize(); print("ms", (time.perf_counter() - t0) / 10 * 1e3)
