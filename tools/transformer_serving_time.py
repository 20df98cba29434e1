"""config 3 throughput: n independent one-clip transformer_lstm requests merged into one pass (TransformerLstm.forward_segments),
in the exact form (every kernel the lone request's: bit-identical results, at most 128 clips a pass) and in the throughput form
(passes of 64 clips or more: large GEMM tiles + the 16-clip persistent stack launch), one pass at a time and - through a
ReasonerServer - with two passes in flight.   python tools/transformer_serving_time.py [heads] [--kernels]"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from objectpermanence_amd import ModelsFactory          # noqa: E402
from objectpermanence_amd.serving import ReasonerServer  # noqa: E402
from synthdata import opnet as synth                     # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
heads = int(args[0]) if args else 4
cfg = {"boxes_features_dim": 256, "num_attention_heads": heads, "num_attention_layers": 2, "num_lstm_layers": 2, "lstm_hidden_dim": 512}
m = ModelsFactory.get_model("transformer_lstm", cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.transformer_lstm_synth_params(cfg).items()})
m = m.eval().to("cuda:0")


def timed(fn, reps=10):
    fn(); fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


base = torch.from_numpy(synth.boxes5(synth.make_batch(0, 64, 300)[0])).cuda()
for n in (1, 16, 32, 64, 128, 256, 512):
    x = torch.cat([base] * ((n + 63) // 64))[:n].contiguous()
    for exact in (True, False):
        if exact and n > m.max_requests_per_pass(1, 300, exact=True):
            continue
        if not exact and n < 64:
            continue
        with torch.no_grad():
            ms = timed(lambda: m.forward_segments(x, n, exact=exact) if n > 1 else m(x))
        print(f"{n:4d} one-clip requests in one pass ({'exact' if exact else 'throughput'} form, engine {getattr(m, 'last_pass_engine', '?')}): "
              f"{ms:7.3f} ms  {n / ms * 1e3:9.1f} clips/s", flush=True)

reqs = [base[i:i + 1].contiguous() for i in range(64)]
for per_pass, exact, streams in ((16, True, 2), (64, False, 1), (64, False, 2), (128, False, 1), (128, False, 2), (256, False, 2), (512, False, 2)):
    server = ReasonerServer(m, "transformer_lstm", max_clips=per_pass, exact=exact, streams=streams)
    total = max(1024, 4 * per_pass)

    def serve():
        with torch.no_grad():
            hs = [server.submit(reqs[i % 64]) for i in range(total)]
            server.flush()
            return [h.result() for h in hs][-1]

    ms = timed(serve, reps=3)
    print(f"server: {per_pass:4d} requests per pass, {streams} pass(es) in flight, {'exact' if exact else 'throughput'}: {total / ms * 1e3:9.1f} clips/s "
          f"({ms / (total / per_pass):.3f} ms per pass)", flush=True)

if "--kernels" in sys.argv:
    # one throughput pass of 128 under the torch profiler: where the time goes
    from torch.profiler import profile, ProfilerActivity
    x = torch.cat([base] * 2).contiguous()
    with torch.no_grad():
        m.forward_segments(x, 128)
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(3):
                m.forward_segments(x, 128)
            torch.cuda.synchronize()
    print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))
