"""Which allocations make the peak of a transformer_lstm training step?  python tools/mem_probe.py B  (prints torch.empty / zeros calls >= 32 MB)"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from objectpermanence_amd import ModelsFactory, FusedAdam
from objectpermanence_amd.training import train_step
from oracle import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
cfg = {"boxes_features_dim": 256, "num_attention_heads": 2, "num_attention_layers": 2, "num_lstm_layers": 2, "lstm_hidden_dim": 512}
m = ModelsFactory.get_model("transformer_lstm", cfg)
m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.transformer_lstm_synth_params(cfg).items()})
m.to("cuda:0").train(True)
opt = FusedAdam(m.parameters(), lr=1e-4)
b, l = synth.make_batch(0, min(B, 4), 300)
x = torch.from_numpy(np.tile(synth.boxes5(b), ((B + 3) // 4, 1, 1, 1))[:B].copy()).cuda()
y = torch.from_numpy(np.tile(l, ((B + 3) // 4, 1, 1))[:B].copy()).cuda()
for name in ("empty", "zeros", "empty_like", "zeros_like"):
    orig = getattr(torch, name)
    def wrap(*a, _o=orig, _n=name, **k):
        t = _o(*a, **k)
        if t.is_cuda and t.numel() * t.element_size() >= 32 << 20:
            fr = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_stack()[-4:-1]]
            print(f"{_n} {t.numel() * t.element_size() / 2**20:8.1f} MB  {' < '.join(reversed(fr))}", flush=True)
        return t
    setattr(torch, name, wrap)
train_step("transformer_lstm", m, opt, x, y)
torch.cuda.synchronize()
print("peak", torch.cuda.max_memory_allocated() / 2**30, "GiB; step 2:")
torch.cuda.reset_peak_memory_stats()
train_step("transformer_lstm", m, opt, x, y)
torch.cuda.synchronize()
print("peak", torch.cuda.max_memory_allocated() / 2**30, "GiB")
