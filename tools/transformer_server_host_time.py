"""Where a served throughput-form pass of transformer_lstm spends its time: host seconds to submit + flush a pass of 256 one-clip
requests (nothing waited for) against the GPU's seconds per pass, and the same pass through forward_segments directly.
python tools/transformer_server_host_time.py [requests_per_pass]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from objectpermanence_amd import ModelsFactory          # noqa: E402
from objectpermanence_amd.serving import ReasonerServer  # noqa: E402
from synthdata import opnet as synth                     # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
cfg = {"boxes_features_dim": 256, "num_attention_heads": 4, "num_attention_layers": 2, "num_lstm_layers": 2, "lstm_hidden_dim": 512}
m = ModelsFactory.get_model("transformer_lstm", cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.transformer_lstm_synth_params(cfg).items()})
m = m.eval().to("cuda:0")
base = torch.from_numpy(synth.boxes5(synth.make_batch(0, 64, 300)[0])).cuda()
reqs = [base[i:i + 1].contiguous() for i in range(64)]
server = ReasonerServer(m, "transformer_lstm", max_clips=n, exact=False)
passes = 8
with torch.no_grad():
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        hs = [server.submit(reqs[i % 64]) for i in range(n * passes)]
        server.flush()
        t1 = time.perf_counter()
        out = [h.result() for h in hs]
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print(f"server, {n} per pass x {passes}: host submit+flush {1e3 * (t1 - t0) / passes:.3f} ms/pass, results {1e3 * (t2 - t1) / passes:.3f} ms/pass, "
              f"all done after {1e3 * (t3 - t0) / passes:.3f} ms/pass -> {n * passes / (t3 - t0):.0f} clips/s", flush=True)
    x = torch.cat([base] * ((n + 63) // 64))[:n].contiguous()
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(passes):
            y = m.forward_segments(x, n, exact=False)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print(f"forward_segments x {passes}: host {1e3 * (t1 - t0) / passes:.3f} ms/pass, done after {1e3 * (t3 - t0) / passes:.3f} ms/pass -> "
              f"{n * passes / (t3 - t0):.0f} clips/s", flush=True)
