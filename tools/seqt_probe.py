"""Probe of the throughput form of the persistent stacked LSTM (csrc/seq_xcdt_kernels.hip): parity against the fp64 oracle
and the 4-clip latency form, then timing over clip counts.  Tools may import oracle/ (never the product).

    python tools/seqt_probe.py [--models baseline_lstm,non_linear_lstm] [--clips 64,128,256,512] [--T 300] [--trace]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from oracle import opnet_oracle as oo, synth  # noqa: E402

REAL = {"baseline_lstm": {"videos_hidden_dim": 512},
        "non_linear_lstm": {"boxes_features_dim": 256, "videos_hidden_dim": 512},
        "transformer_lstm": {"boxes_features_dim": 256, "num_attention_heads": 2, "num_attention_layers": 2,
                             "num_lstm_layers": 2, "lstm_hidden_dim": 512}}
PARAMS = {"baseline_lstm": synth.baseline_lstm_synth_params, "non_linear_lstm": synth.non_linear_lstm_synth_params,
          "transformer_lstm": synth.transformer_lstm_synth_params}
ORACLE = {"baseline_lstm": lambda x, p, cfg: oo.baseline_lstm_forward(x, p),
          "non_linear_lstm": lambda x, p, cfg: oo.non_linear_lstm_forward(x, p),
          "transformer_lstm": lambda x, p, cfg: oo.transformer_lstm_forward(x, p, cfg)}
FLOP = {"baseline_lstm": 2 * 2048 * (80 + 512), "non_linear_lstm": 2 * 2048 * (512 + 1024),      # recurrence only, per clip-frame
        "transformer_lstm": 2 * 2048 * (512 + 1024)}


def model(name, mode):
    from objectpermanence_amd import ModelsFactory
    cfg = REAL[name]
    m = ModelsFactory.get_model(name, cfg)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in PARAMS[name](cfg).items()})
    m._runner.use_xcdt = mode
    return m.eval().to("cuda:0")


def run(m, x):
    with torch.no_grad():
        y = m(x)
    torch.cuda.synchronize()
    return y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", default="baseline_lstm,non_linear_lstm")
    ap.add_argument("--clips", default="64,128,256,512")
    ap.add_argument("--T", type=int, default=300)
    ap.add_argument("--parity", default="1,17,40,100")
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    from objectpermanence_amd import _lib
    lib = _lib.load()
    for name in a.models.split(","):
        cfg = REAL[name]
        mt, mx = model(name, "1"), model(name, "0")
        for B in [int(v) for v in a.parity.split(",") if v]:
            T = 23 if B > 40 else 9
            boxes, _ = synth.make_batch(500, B, T)
            x = synth.boxes5(boxes)
            xd = torch.from_numpy(x).cuda()
            y = run(mt, xd).cpu().numpy()
            bad = mt._runner._monitor.verify()
            y_ref = ORACLE[name](x, PARAMS[name](cfg), cfg)
            y4 = run(mx, xd).cpu().numpy()
            y2 = run(mt, xd).cpu().numpy()
            print(f"{name} B={B} T={T}: launches={mt._runner.xcdt_launches} aborted={bad} max|y-oracle|={np.abs(y - y_ref).max():.3e} "
                  f"max|y-latency form|={np.abs(y - y4).max():.3e} deterministic={np.array_equal(y, y2)} finite={np.isfinite(y).all()}", flush=True)
        for B in [int(v) for v in a.clips.split(",") if v]:
            boxes, _ = synth.make_batch(0, min(B, 64), a.T)
            x = np.concatenate([synth.boxes5(boxes)] * ((B + 63) // 64))[:B]
            xd = torch.from_numpy(x).cuda()
            for tag, m in (("throughput", mt), ("latency", mx)):
                if tag == "latency" and B > 256:
                    continue
                run(m, xd)
                lib.opnet_xcd_profile(1)
                t0 = time.perf_counter()
                for _ in range(a.reps):
                    run(m, xd)
                dt = (time.perf_counter() - t0) / a.reps
                import ctypes
                ms, nl = ctypes.c_double(0), ctypes.c_int(0)
                lib.opnet_kernel_profile_read(3 if tag == "throughput" else 1, ctypes.byref(ms), ctypes.byref(nl))
                lib.opnet_xcd_profile(0)
                kms = ms.value / max(nl.value, 1) * (nl.value / a.reps)
                tf = FLOP[name] * B * a.T / (kms * 1e-3) / 1e12 if kms > 0 else 0.0
                print(f"{name} {tag:10s} B={B:4d} T={a.T}: forward {dt * 1e3:7.3f} ms  {B / dt / 1e3:7.1f} k clips/s   persistent kernel "
                      f"{kms:7.3f} ms = {tf:6.1f} TF ({tf / 157.3:.3f} of fp32 MFMA peak)", flush=True)
        if a.trace:
            B = int(a.clips.split(",")[-1])
            T = a.T
            boxes, _ = synth.make_batch(0, min(B, 64), T)
            x = np.concatenate([synth.boxes5(boxes)] * ((B + 63) // 64))[:B]
            xd = torch.from_numpy(x).cuda()
            buf = torch.zeros(2 * (T + 1) * 8 * 8, dtype=torch.int64, device="cuda:0")
            lib.opseq_xcdt_set_trace(buf.data_ptr())
            run(mt, xd)
            lib.opseq_xcdt_set_trace(None)
            tr = buf.cpu().numpy().reshape(2, (T + 1) * 8, 8)
            for blk in range(2):
                t = tr[blk]
                ok = t[:, 0] > 0
                ph = t[ok]
                if len(ph) < 40:
                    continue
                mid = ph[len(ph) // 4: 3 * len(ph) // 4]
                period = np.diff(mid[:, 0]).mean()
                prod = (mid[:, 1] - mid[:, 0]).mean()
                fin = (mid[:, 6] - mid[:, 2]).mean()
                cell = (mid[:, 4] - mid[:, 3]).mean()
                pub = (mid[:, 5] - mid[:, 4]).mean()
                gat = (mid[:, 6] - mid[:, 5]).mean()
                print(f"{name} trace block {blk}: phases {len(ph)}  period {period:.0f} cycles  products {prod:.0f}  finish wave 0: barrier->done {fin:.0f} "
                      f"(cell {cell:.0f}, publish {pub:.0f}, poll+gather {gat:.0f})", flush=True)


if __name__ == "__main__":
    main()
