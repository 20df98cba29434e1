"""Randomised sweep of the throughput form of the persistent stacked LSTM (csrc/seq_xcdt_kernels.hip) against the fp64 oracle, with
the chip UNEVENLY loaded by a second stream (cdna_hip_programming.md Guideline 16: idle chips and uniform load hide stale
hand-offs): fresh weights per case, random clip and frame counts, both hand-off protocols, every output word checked, every case
run twice (bit-identical or it fails).  Not part of the test suite; on the GPU box:  CASES=40 SEED=0 python tools/fuzz_seqt.py"""
import os
import sys
import threading

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from objectpermanence_amd import ModelsFactory
from oracle import opnet_oracle as oo, synth

CFGS = {"baseline_lstm": {"videos_hidden_dim": 512},
        "non_linear_lstm": {"boxes_features_dim": 16, "videos_hidden_dim": 512},      # 15 x 16 = 240 features: hoisted, cheap oracle
        "transformer_lstm": {"boxes_features_dim": 256, "num_attention_heads": 2, "num_attention_layers": 1, "num_lstm_layers": 2,
                             "lstm_hidden_dim": 512}}
PARAMS = {"baseline_lstm": synth.baseline_lstm_synth_params, "non_linear_lstm": synth.non_linear_lstm_synth_params,
          "transformer_lstm": synth.transformer_lstm_synth_params}
ORACLE = {"baseline_lstm": lambda x, p, cfg: oo.baseline_lstm_forward(x, p),
          "non_linear_lstm": lambda x, p, cfg: oo.non_linear_lstm_forward(x, p),
          "transformer_lstm": lambda x, p, cfg: oo.transformer_lstm_forward(x, p, cfg)}
rng = np.random.default_rng(int(os.environ.get("SEED", "0")))
worst = 0.0

# uneven load: a second stream keeps a varying part of the chip busy with GEMMs of random sizes while the launches under test run
stop = threading.Event()


def disturb():
    st = torch.cuda.Stream()
    g = torch.Generator().manual_seed(1)
    mats = [torch.randn(int(n), int(n), device="cuda:0") for n in (256, 512, 1024, 2048)]
    with torch.cuda.stream(st):
        while not stop.is_set():
            a = mats[int(torch.randint(0, 4, (1,), generator=g))]
            for _ in range(int(torch.randint(1, 6, (1,), generator=g))):
                a = (a @ a) * 1e-3
            st.synchronize()


th = threading.Thread(target=disturb, daemon=True)
if os.environ.get("DISTURB", "1") == "1":
    th.start()
for case in range(int(os.environ.get("CASES", "40"))):
    name = ["baseline_lstm", "non_linear_lstm", "transformer_lstm"][case % 3]
    cfg = CFGS[name]
    B = int(rng.choice([1, 15, 16, 17, 40, 64, 100, 128, 129, 200, 256, 300, 384, 512, 530] if name != "transformer_lstm" else [1, 16, 40, 70, 150]))
    T = int(rng.choice([1, 2, 3, 5, 9, 17, 33]))
    os.environ["OPNET_XCD_SAFE"] = str(int(rng.integers(0, 2)))
    p = PARAMS[name](cfg, salt=case) if name != "transformer_lstm" else PARAMS[name](cfg, salt=case)
    boxes, _ = synth.make_batch(3000 + case, min(B, 64), T)
    boxes = np.tile(boxes, ((B + boxes.shape[0] - 1) // boxes.shape[0], 1, 1, 1))[:B].copy()
    boxes[:, :, :, :4] += (np.arange(B, dtype=np.float32) % 11)[:, None, None, None] * 1e-3          # no two clips alike
    x = synth.boxes5(boxes)
    m = ModelsFactory.get_model(name, cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in p.items()})
    m.eval().to("cuda:0")
    m._runner.use_xcdt = "1"
    xd = torch.from_numpy(x).cuda()
    with torch.no_grad():
        y = m(xd).cpu().numpy()
        y2 = m(xd).cpu().numpy()
    assert m._runner._monitor.verify() == 0, "a persistent launch aborted"
    ref = ORACLE[name](x, p, cfg)
    err = float(np.abs(y - ref).max())
    worst = max(worst, err)
    ok = err < 3e-5 and np.array_equal(y, y2) and np.isfinite(y).all()
    print(f"case {case}: {name} B={B} T={T} safe={os.environ['OPNET_XCD_SAFE']} launches={m._runner.xcdt_launches} |dy| {err:.2e} "
          f"deterministic={np.array_equal(y, y2)} {'ok' if ok else 'FAIL'}", flush=True)
    assert ok
stop.set()
if th.is_alive():
    th.join()
torch.cuda.synchronize()
print(f"WORST |dy| {worst:.3e}")
