"""Timing and in-kernel timeline of the 4-clip persistent training step (csrc/opnet_xcd4_kernels.hip).
    python tools/xcd4_probe.py [--batches 32] [--frames 300] [--reps 5]"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectpermanence_amd import ModelsFactory, _lib, l1_mean  # noqa: E402
from synthdata import opnet as synth  # noqa: E402

CFG = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}
NAMES = ["products", "barrier 1", "cell + exchange store", "history stores", "gather (poll until published)", "drain", "barrier 2", "loop"]


def timed(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="32")
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    m = ModelsFactory.get_model("opnet", CFG)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in synth.opnet_synth_params(CFG).items()})
    m = m.to(dev).train(True)
    T = args.frames
    for B in [int(v) for v in args.batches.split(",")]:
        boxes, labels = synth.make_batch(0, B, T)
        x, lab = torch.from_numpy(boxes).to(dev), torch.from_numpy(labels).to(dev)

        def fwd():
            return m(x)

        def step():
            m.zero_grad(set_to_none=True)
            y, _ = m(x)
            l1_mean(y, lab).backward()

        res = {}
        for mode in ("1", "0"):
            os.environ["OPNET_XCD4"] = mode
            res[mode] = (timed(fwd, args.reps), timed(step, args.reps))
        os.environ["OPNET_XCD4"] = "1"
        print(f"B={B} T={T}: forward persistent {res['1'][0]:.3f} ms / chain {res['0'][0]:.3f} ms | fwd+loss+bwd persistent "
              f"{res['1'][1]:.3f} ms / chain {res['0'][1]:.3f} ms", flush=True)
        st = (ctypes.c_uint * 4)()
        lib.opnet_xcd4_last_status(st)
        print(f"  status of the last persistent launch: {list(st)}", flush=True)
        ng = (B + 31) // 32
        tr = torch.zeros((T + 3) * ng * 8, dtype=torch.int64, device=dev)
        lib.opnet_xcd4_set_trace(tr.data_ptr())
        fwd()
        torch.cuda.synchronize()
        lib.opnet_xcd4_set_trace(None)
        t = tr.cpu().numpy().reshape(-1, 8)[:(T + 2) * ng]
        ph = t[len(t) // 3: 2 * len(t) // 3]
        med = lambda v: float(np.median(v))
        parts = [med(ph[:, i + 1] - ph[:, i]) for i in range(7)] + [med(ph[1:, 0] - ph[:-1, 7])]
        print(f"  forward, block 0 wave 0, cycles (median): period {med(np.diff(ph[:, 0])):.0f} | "
              + ", ".join(f"{n} {v:.0f}" for n, v in zip(NAMES, parts)), flush=True)
        tr.zero_()
        lib.opnet_xcd4_set_trace(tr.data_ptr())
        step()          # the backward kernel stamps the same buffer after the forward
        torch.cuda.synchronize()
        lib.opnet_xcd4_set_trace(None)
        t = tr.cpu().numpy().reshape(-1, 8)
        t = t[:(T + 3) * ng]
        ph = t[len(t) // 3: 2 * len(t) // 3]
        BN = ["cells (wave 0: chunk sum, cell)", "barrier 1", "products + partial stores", "gather (poll until published)",
              "drain", "barrier 2"]
        parts = [med(ph[:, i + 1] - ph[:, i]) for i in range(6)] + [med(ph[1:, 0] - ph[:-1, 6])]
        print(f"  backward, block 0 wave 0, cycles (median; each stamp costs ~200): period {med(np.diff(ph[:, 0])):.0f} | "
              + ", ".join(f"{n} {v:.0f}" for n, v in zip(BN + ["loop"], parts)), flush=True)

        sx = lambda v: ((v.astype(np.int64) & 0xffff) ^ 0x8000) - 0x8000
        d = ph[:, 7].astype(np.int64)
        print("  backward: arrival at barrier 2 relative to wave 0, cycles (median): wave 1 %+.0f, wave 2 %+.0f, wave 3 %+.0f" % (
            med(sx(d)), med(sx(d >> 16)), med(sx(d >> 32))), flush=True)

if __name__ == "__main__":
    main()
