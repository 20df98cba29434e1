#!/usr/bin/env python3
"""Timing of the per-XCD persistent OPNet forward (opnet_xcd_forward_f32) against the step-launch form.
    python tools/xcd_probe.py [--batches 32,64,128,256,512] [--frames 300] [--trace]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectpermanence_amd import ModelsFactory, _lib  # noqa: E402
from synthdata import opnet as synth  # noqa: E402

CFG = {"object_to_track_pred_dim": 15, "object_to_track_hidden_dim": 256, "videos_hidden_dim": 512}


def timed(fn, reps):
    import gc
    fn(); torch.cuda.synchronize()
    gc.collect(); gc.disable()                  # a collection inside the loop idles the GPU for milliseconds (seen: 320 clips 4.9 instead of 2.9 ms)
    try:
        return _timed(fn, reps)
    finally:
        gc.enable()


def _timed(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="32,64,128,256,512,1024")
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--trace", action="store_true")
    ap.add_argument("--no-chain", action="store_true")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    m = ModelsFactory.get_model("opnet", CFG)
    params = synth.opnet_synth_params(CFG)
    m.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in params.items()})
    m = m.eval().to(dev)
    T = args.frames
    for B in [int(b) for b in args.batches.split(",")]:
        boxes, _ = synth.make_batch(0, min(B, 64), T)
        boxes = np.tile(boxes, ((B + boxes.shape[0] - 1) // boxes.shape[0], 1, 1, 1))[:B]
        x = torch.from_numpy(boxes).to(dev)
        res = {}
        with torch.no_grad():
            m.use_xcd = "1"
            y1, l1 = m(x)
            ms_x = timed(lambda: m(x), args.reps)
            st = m.xcd_status()
            res["xcd"] = ms_x
            if not args.no_chain:
                m.use_xcd = "0"
                y0, l0 = m(x)
                res["chain"] = timed(lambda: m(x), args.reps)
                res["max|dy|"] = float((y1 - y0).abs().max())
                res["max|dlogits|"] = float((l1 - l0).abs().max())
        flops = B * 852.7e6
        print(f"B={B:5d} T={T}: xcd {ms_x:8.3f} ms = {B / ms_x * 1e3:9.0f} clips/s ({flops / ms_x * 1e-9 / 157.3:.3f} of fp32 MFMA peak)"
              + (f" | chain {res['chain']:8.3f} ms = {B / res['chain'] * 1e3:9.0f} clips/s | max|dy| {res['max|dy|']:.2e} max|dlg| {res['max|dlogits|']:.2e}"
                 if "chain" in res else "") + f" | status {list(st.values())[-1]}", flush=True)
    if args.trace:
        lib = _lib.load()
        for B in (128, 256, 384, 512, 640):
            ng = (B + 127) // 128
            tr = torch.zeros((T + 4) * ng * 8, dtype=torch.int64, device=dev)
            lib.opnet_xcd_set_trace(tr.data_ptr())
            boxes, _ = synth.make_batch(0, 64, T)
            x = torch.from_numpy(np.tile(boxes, ((B + 63) // 64, 1, 1, 1))[:B]).to(dev)
            m.use_xcd = "1"
            with torch.no_grad():
                m(x)
            torch.cuda.synchronize()
            lib.opnet_xcd_set_trace(None)
            t = tr.cpu().numpy().reshape(-1, 8)
            ph = t[len(t) // 3: 2 * len(t) // 3]
            med = lambda v: float(np.median(v))
            print("B=%d (%d groups per XCD), cycles (median): period %.0f | product wave: products %.0f, hand-off + barrier wait %.0f | "
                  "finish wave after the barrier: early-gather %.0f, head %.0f, cells %.0f, drain+flag %.0f, late poll+gather+land %.0f, "
                  "to next barrier %.0f" % (
                      B, ng, med(np.diff(ph[:, 0])), med(ph[:, 1] - ph[:, 0]), med(ph[1:, 0] - ph[:-1, 1]),
                      med(ph[:, 3] - ph[:, 2]), med(ph[:, 4] - ph[:, 3]), med(ph[:, 5] - ph[:, 4]), med(ph[:, 6] - ph[:, 5]),
                      med(ph[:, 7] - ph[:, 6]), med(ph[1:, 2] - ph[:-1, 7])))
            # block 0 = CU 0 of XCD 0: its phases by role (selection head / output head of the phase's step on this CU / neither)
            fp = np.arange(len(t))[len(t) // 3: 2 * len(t) // 3 - 1]
            gi, st_ = fp % ng, fp // ng
            role = np.where(((st_ + 11 * gi) & 31) == 0, 1, np.where(((st_ + 11 * gi + 16) & 31) == 0, 2, 0))
            per, prod = t[fp + 1, 0] - t[fp, 0], t[fp, 1] - t[fp, 0]
            print("   mean period %.0f; by role (plain / sel head / out head): period %s, products %s" % (
                per.mean(), [round(float(per[role == r].mean())) for r in (0, 1, 2)],
                [round(float(prod[role == r].mean())) for r in (0, 1, 2)]))


if __name__ == "__main__":
    main()
