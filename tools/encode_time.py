"""Input-encode throughput: clips/s of the native encoder from preloaded flat arrays, and the per-clip host cost
of a dataset sample (<video>.pkl + <video>_bb.json -> tensors) with the native and the numpy encoder.
    python tools/encode_time.py [n_clips]"""
import json
import os
import pickle
import sys
import tempfile
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectpermanence_amd import datasets as D  # noqa: E402
from synthdata import opnet as synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
raws = [synth.make_raw_video(i, "plain") for i in range(16)]
flat = [D.flatten_detections(bb, lab) for bb, lab, _ in raws]
C = np.concatenate([flat[k % 16][0] for k in range(n)])
I = np.concatenate([flat[k % 16][1] for k in range(n)])
F = np.concatenate([flat[k % 16][2] for k in range(n)])
out, idx = np.empty((n, 300, 15, 6), np.float32), np.empty((n, 300), np.int64)
D.encode_clips_native(C, I, F, n, 300, 6, out=out, idx=idx)
best = float("inf")
for _ in range(5):
    t0 = time.perf_counter()
    D.encode_clips_native(C, I, F, n, 300, 6, out=out, idx=idx)
    best = min(best, time.perf_counter() - t0)
print(f"native encoder, {n} clips from preloaded arrays: {n / best:.0f} clips/s ({best / n * 1e6:.1f} us/clip, one thread)")
t0 = time.perf_counter()
for k in range(32):
    bb, lab, _ = raws[k % 16]
    b = D.encode_boxes(bb, lab, 6)
    D.index_to_track(b)
dt = (time.perf_counter() - t0) / 32
print(f"numpy statement (encode_boxes + index_to_track): {1 / dt:.0f} clips/s ({dt * 1e3:.2f} ms/clip)")
with tempfile.TemporaryDirectory() as tmp:
    s, l = os.path.join(tmp, "s"), os.path.join(tmp, "l")
    os.mkdir(s); os.mkdir(l)
    for k in range(64):
        bb, lab, gt = raws[k % 16]
        pickle.dump({"bb": bb, "labels": lab}, open(os.path.join(s, f"v{k:03d}.pkl"), "wb"), pickle.HIGHEST_PROTOCOL)
        json.dump(gt, open(os.path.join(l, f"v{k:03d}_bb.json"), "w"))
    for native in ("1", "0"):
        os.environ["OPNET_NATIVE_ENCODE"] = native
        ds = D.Cater6TracksForObjectsInferenceDataset(s, l)
        ds[0]
        t0 = time.perf_counter()
        for k in range(len(ds)):
            ds[k]
        dt = (time.perf_counter() - t0) / len(ds)
        print(f"dataset sample (pickle.load + json + encode + tensors), native={native}: {dt * 1e3:.2f} ms/clip = {1 / dt:.0f} clips/s per worker")
