import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from objectpermanence_amd import ModelsFactory
from oracle import synth
def bench(name, cfg, pf, B, feat, n=5):
    m = ModelsFactory.get_model(name, cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pf(cfg).items()})
    m.eval().to("cuda:0")
    b = synth.make_batch(0, min(B, 4), 300)[0]
    x = torch.from_numpy(np.tile(b, ((B + 3) // 4, 1, 1, 1))[:B][..., :feat].copy()).cuda()
    with torch.no_grad():
        for _ in range(2): m(x)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): m(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{name} B={B}: {dt*1e3:.3f} ms/forward  {B/dt:.0f} clips/s", flush=True)
tcfg = {"boxes_features_dim": 256, "num_attention_heads": 2, "num_attention_layers": 2, "num_lstm_layers": 2, "lstm_hidden_dim": 512}
for B in (1, 4, 16, 32, 64): bench("transformer_lstm", tcfg, synth.transformer_lstm_synth_params, B, 5)
t4 = dict(tcfg); t4["num_attention_heads"] = 4
bench("transformer_lstm", t4, synth.transformer_lstm_synth_params, 1, 5)
bench("baseline_lstm", {"videos_hidden_dim": 512}, synth.baseline_lstm_synth_params, 32, 5)
bench("non_linear_lstm", {"boxes_features_dim": 256, "videos_hidden_dim": 512}, synth.non_linear_lstm_synth_params, 32, 5)
bench("opnet_lstm_mlp", {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}, synth.opnet_lstm_mlp_synth_params, 32, 6)
