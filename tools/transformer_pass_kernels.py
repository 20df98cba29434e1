"""Every kernel of ONE throughput-form pass of n one-clip transformer_lstm requests, in launch order, with its duration
(torch profiler on the third of three passes).   python tools/transformer_pass_kernels.py [n = 256] [heads = 4]"""
import sys

import torch
from torch.profiler import profile, ProfilerActivity

sys.path.insert(0, ".")
from objectpermanence_amd import ModelsFactory          # noqa: E402
from synthdata import opnet as synth                     # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
heads = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cfg = {"boxes_features_dim": 256, "num_attention_heads": heads, "num_attention_layers": 2, "num_lstm_layers": 2, "lstm_hidden_dim": 512}
m = ModelsFactory.get_model("transformer_lstm", cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.transformer_lstm_synth_params(cfg).items()})
m = m.eval().to("cuda:0")
base = torch.from_numpy(synth.boxes5(synth.make_batch(0, 64, 300)[0])).cuda()
x = torch.cat([base] * ((n + 63) // 64))[:n].contiguous()
with torch.no_grad():
    for _ in range(3):
        m.forward_segments(x, n)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CUDA]) as prof:
        m.forward_segments(x, n)
        torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type is not None and "DeviceType.CUDA" in str(e.device_type)]
evs.sort(key=lambda e: e.time_range.start)
t0 = evs[0].time_range.start
total = 0.0
for e in evs:
    d = e.time_range.end - e.time_range.start
    total += d
    print(f"{(e.time_range.start - t0):10.1f} us  {d:9.1f} us  {e.name[:110]}")
print(f"sum of kernel time {total / 1e3:.3f} ms, span {(evs[-1].time_range.end - t0) / 1e3:.3f} ms, {len(evs)} kernels, n = {n}")
