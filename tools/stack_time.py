"""Time one forward of the stacked-LSTM reasoners on both engines (persistent launch / launch chain).
    python tools/stack_time.py [--reps 20]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectpermanence_amd import ModelsFactory  # noqa: E402
from synthdata import opnet as synth  # noqa: E402

CFG = {"baseline_lstm": {"videos_hidden_dim": 512},
       "non_linear_lstm": {"boxes_features_dim": 256, "videos_hidden_dim": 512},
       "transformer_lstm": {"boxes_features_dim": 256, "num_attention_heads": 4, "num_attention_layers": 2,
                            "num_lstm_layers": 2, "lstm_hidden_dim": 512}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--T", type=int, default=300)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    for name, batches in (("baseline_lstm", (1, 16, 32, 64)), ("non_linear_lstm", (1, 16, 32)), ("transformer_lstm", (1, 4, 16, 32))):
        for B in batches:
            boxes, _ = synth.make_batch(0, min(B, 32), args.T)
            x = torch.from_numpy(np.tile(synth.boxes5(boxes), ((B + 31) // 32, 1, 1, 1))[:B]).to(dev)
            row = []
            for engine in ("auto", "chain"):
                torch.manual_seed(0)
                m = ModelsFactory.get_model(name, CFG[name]).eval().to(dev)
                if engine == "chain":
                    m._runner.use_xcd = "0"
                with torch.no_grad():
                    for _ in range(3):
                        m(x)
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.reps):
                        m(x)
                    e1.record()
                    torch.cuda.synchronize()
                assert m._runner._monitor.verify() == 0
                row.append((engine, e0.elapsed_time(e1) / args.reps, m._runner.xcd_launches))
            print(f"{name:18s} B={B:3d} T={args.T}: " + "  ".join(f"{e}: {ms:.3f} ms ({B / ms * 1e3:.0f} clips/s, persistent launches {n})"
                                                              for e, ms, n in row), flush=True)


if __name__ == "__main__":
    main()
