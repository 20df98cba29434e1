"""config 3 throughput: n independent one-clip transformer_lstm requests, alone one after the other vs merged into one pass
(TransformerLstm.forward_segments).  python tools/transformer_serving_time.py [heads]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from objectpermanence_amd import ModelsFactory          # noqa: E402
from synthdata import opnet as synth                     # noqa: E402

heads = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = {"boxes_features_dim": 256, "num_attention_heads": heads, "num_attention_layers": 2, "num_lstm_layers": 2, "lstm_hidden_dim": 512}
m = ModelsFactory.get_model("transformer_lstm", cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.transformer_lstm_synth_params(cfg).items()})
m = m.eval().to("cuda:0")


def timed(fn, reps=20):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


for n in (1, 4, 8, 16, 32, 64, 128):
    x = torch.from_numpy(synth.boxes5(synth.make_batch(0, n, 300)[0])).cuda()
    with torch.no_grad():
        ms = timed(lambda: m.forward_segments(x, n) if n > 1 else m(x))
    print(f"{n:4d} one-clip requests in one pass: {ms:7.3f} ms  {n / ms * 1e3:9.1f} clips/s", flush=True)
