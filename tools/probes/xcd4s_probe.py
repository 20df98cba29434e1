"""In-kernel timeline of the split 4-clip persistent forward (csrc/opnet_xcd4s_kernels.hip): s_memtime stamps of CU 0 of XCD 0,
both roles, every wave.    python tools/xcd4s_probe.py [--batch 32] [--frames 300] [--train 1]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectpermanence_amd import ModelsFactory, _lib  # noqa: E402
from synthdata import opnet as synth  # noqa: E402

CFG = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}
NAMES = ["finish of the previous phase (w0: LSTM2 cell, w3: LSTM1 cell, w2: head)", "asks + x side", "wait for the wave's K quarter of h2",
         "LDS + LSTM2's 128 MFMAs", "wait for h1 + LSTM1 / head MFMAs + partial sums", "barrier", "loop"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--train", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    m = ModelsFactory.get_model("opnet", CFG)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.opnet_synth_params(CFG).items()})
    m = m.to(dev).train(bool(args.train))
    B, T = args.batch, args.frames
    x = torch.from_numpy(synth.make_batch(0, B, T)[0]).to(dev)
    run = (lambda: m(x)) if args.train else (lambda: torch.no_grad().__enter__() or m(x))
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"B={B} T={T} train={args.train}: forward (all launches) {np.median(ts):.3f} ms, OPNET_X4_SPLIT={os.environ.get('OPNET_X4_SPLIT', '1')}")
    ng = (B + 31) // 32
    nph = (T + 3) * ng
    tr = torch.zeros(4 * nph * 8, dtype=torch.int64, device=dev)
    lib.opnet_xcd4_set_trace(tr.data_ptr())
    run()
    torch.cuda.synchronize()
    lib.opnet_xcd4_set_trace(None)
    t = tr.cpu().numpy().reshape(4, nph, 8)
    med = lambda v: float(np.median(v))
    lo, hi = nph // 3, 2 * nph // 3
    for w in range(4):
        ph = t[w, lo:hi]
        period = med(np.diff(ph[:, 0]))
        parts = [med(ph[:, i + 1] - ph[:, i]) for i in range(6)] + [med(ph[1:, 0] - ph[:-1, 6])]
        print(f"  wave {w}: period {period:.0f} cycles | " + ", ".join(f"{n} {v:.0f}" for n, v in zip(NAMES, parts)))
    base = t[0, lo, 0]
    print("  raw stamps of three consecutive phases (cycles from wave 0's first):")
    for w in range(4):
        for p in range(lo, lo + 3):
            print(f"    wave {w} phase {p}: " + " ".join(f"{int(v - base):6d}" for v in t[w, p, :7]))


if __name__ == "__main__":
    main()
