"""Nothing but throughput-form passes of n one-clip transformer_lstm requests (for a profiler: every kernel it sees belongs to such a pass).
    python tools/transformer_pass_only.py [n = 256] [passes = 6] [heads = 4]"""
import sys

import torch

sys.path.insert(0, ".")
from objectpermanence_amd import ModelsFactory          # noqa: E402
from synthdata import opnet as synth                     # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 6
heads = int(sys.argv[3]) if len(sys.argv) > 3 else 4
cfg = {"boxes_features_dim": 256, "num_attention_heads": heads, "num_attention_layers": 2, "num_lstm_layers": 2, "lstm_hidden_dim": 512}
m = ModelsFactory.get_model("transformer_lstm", cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.transformer_lstm_synth_params(cfg).items()})
m = m.eval().to("cuda:0")
base = torch.from_numpy(synth.boxes5(synth.make_batch(0, 64, 300)[0])).cuda()
x = torch.cat([base] * ((n + 63) // 64))[:n].contiguous()
with torch.no_grad():
    for _ in range(passes):
        y = m.forward_segments(x, n)
torch.cuda.synchronize()
assert m._runner._monitor.verify() == 0 and bool(torch.isfinite(y).all())
print(f"{passes} passes of {n} one-clip requests, {heads} heads: done")
