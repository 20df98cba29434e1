#!/usr/bin/env python3
"""Randomised check of objectpermanence_amd.datasets.encode_boxes / index_to_track AND of the native encoder
(encode_clips_native -> csrc/encode_host.cpp) against the reference's own
_normalize_and_pad_predictions / _get_closest_object_to_track_vector (baselines/datasets.py:130-257, 265-416).
Runs only where /root/reference exists (build container); test infrastructure, never imported by the product.

    python oracle/fuzz_datasets.py [n_cases]
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
REF = os.environ.get("OPNET_REFERENCE", "/root/reference")


def main(n_cases: int = 300) -> int:
    sys.path.insert(0, REF)
    if not hasattr(np, "int"):
        np.int = int
    if not hasattr(np, "bool"):
        np.bool = bool
    from baselines import datasets as rd
    from objectpermanence_amd.datasets import encode_boxes, encode_clips_native, flatten_detections, index_to_track
    from objectpermanence_amd.object_indices import CONE_IDS
    cones = sorted(CONE_IDS)
    others = [i for i in range(193) if i not in CONE_IDS and i != 140]
    bad = 0
    for case in range(n_cases):
        rng = np.random.default_rng(case)
        n_obj = int(rng.integers(1, 20))
        ids = list(rng.choice(cones, size=int(rng.integers(0, min(n_obj, 6) + 1)), replace=False))
        ids += list(rng.choice(others, size=max(n_obj - len(ids) - 1, 0), replace=False))
        if rng.random() < 0.8:
            ids.append(140)
        ids = np.array(ids, dtype=np.int64)
        T = int(rng.integers(1, 40))
        p_vis = rng.choice([0.1, 0.5, 0.9])
        bb, lab = [], []
        for t in range(T):
            vis = np.flatnonzero(rng.random(len(ids)) < p_vis) if rng.random() > 0.1 else np.zeros(0, dtype=np.int64)
            if len(vis) > 1 and rng.random() < 0.3:
                vis = np.concatenate([vis, rng.choice(vis, size=2)])
            vis = vis[rng.permutation(len(vis))]
            lab.append(ids[vis])
            x1 = rng.integers(0, 300, size=len(vis)); y1 = rng.integers(0, 220, size=len(vis))
            bb.append(np.stack([x1, y1, x1 + rng.integers(8, 20, size=len(vis)), y1 + rng.integers(8, 20, size=len(vis))], axis=1).reshape(-1, 4))
        for tracks, cls in ((6, rd.CaterAbstract6TracksForObjectsDataset), (5, rd.CaterAbstract5TracksForObjectsDataset)):
            ds = cls.__new__(cls)
            cls.__init__(ds, "/nonexistent", "/nonexistent")
            ref = np.array(ds._normalize_and_pad_predictions(bb, lab))
            mine = encode_boxes(bb, lab, tracks)
            ok = ref.shape == mine.shape and np.array_equal(ref, mine)
            if ok:
                ok = list(ds._get_closest_object_to_track_vector(list(ref))) == index_to_track(mine)
            if ok:      # the native encoder (csrc/encode_host.cpp) against the reference's own outputs, fp32 cast included
                c, i, f = flatten_detections(bb, lab)
                nb, ni = encode_clips_native(c, i, f, 1, T, tracks)
                ok = np.array_equal(nb[0], ref.astype(np.float32)) and \
                    list(ni[0]) == list(ds._get_closest_object_to_track_vector(list(ref)))
            if not ok:
                bad += 1
                print("MISMATCH case", case, "tracks", tracks)
    print(f"{n_cases} cases x 2 encoders: {bad} mismatches")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(int(sys.argv[1]) if len(sys.argv) > 1 else 300) else 0)
