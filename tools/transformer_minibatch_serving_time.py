"""The reference's inference call pattern for transformer_lstm: DataLoader minibatches of 16 clips (configs/inference_config.json),
each ONE coupled request (attention over S = 16 x 300 tokens), merged by the server as segments.  python tools/transformer_minibatch_serving_time.py"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from objectpermanence_amd import ModelsFactory          # noqa: E402
from synthdata import opnet as synth                     # noqa: E402

heads = 2
cfg = {"boxes_features_dim": 256, "num_attention_heads": heads, "num_attention_layers": 2, "num_lstm_layers": 2, "lstm_hidden_dim": 512}
m = ModelsFactory.get_model("transformer_lstm", cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.transformer_lstm_synth_params(cfg).items()})
m = m.eval().to("cuda:0")
base = torch.from_numpy(synth.boxes5(synth.make_batch(0, 64, 300)[0])).cuda()


def timed(fn, reps=5):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


for b in (16,):
    for n in (1, 2, 4, 8, 16, 32):
        if n > m.max_requests_per_pass(b, 300):
            continue
        x = torch.cat([base] * ((n * b + 63) // 64))[:n * b].contiguous()
        with torch.no_grad():
            ms = timed(lambda: m.forward_segments(x, n) if n > 1 else m(x))
        print(f"{n:3d} requests of {b} clips in one pass (engine {m.last_pass_engine}): {ms:8.3f} ms  {n * b / ms * 1e3:9.1f} clips/s", flush=True)
from torch.profiler import profile, ProfilerActivity
n, b = 16, 16
x = torch.cat([base] * 4)[:n * b].contiguous()
with torch.no_grad():
    m.forward_segments(x, n)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        m.forward_segments(x, n)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=12, max_name_column_width=50))
