#!/usr/bin/env python3
"""Reduce rocprofv3 PMC passes (one counter set per pass, collected with --pmc only - never together with trace domains) to
per-kernel numbers.

    python tools/pmc_reduce.py traffic <fetch_dir> <write_dir> <kernel-substring> [<algorithmic bytes per launch>]
        HBM-side bytes per launch of the kernel: FETCH_SIZE x 2 (MI355X_MICROARCH.md HBM section: on gfx950 the counter
        reports 1/2 of the bytes of 16-B/lane coalesced reads) + WRITE_SIZE, counter unit KB
    python tools/pmc_reduce.py shapes <fetch_dir> <write_dir> <clips,clips,...> [<json to merge into>]
        the same per launch SHAPE of bench.py's kernels (opnet_xcd_forward<true>, opnet_xcd_pack_input, the output-head tails): the two
        passes run the same command, so launch i of a kernel is the same launch in both; a launch of opnet_xcd_forward is assigned to
        the candidate clip count whose compulsory read (115.2 KB of packed input per clip + the weights once per XCD) is nearest
        to its corrected FETCH_SIZE, the other kernels follow the forward they belong to.  Prints profiles/r3_pmc_traffic.json.
    python tools/pmc_reduce.py mfma <dir> [<dir> ...]
        MFMA utilisation per kernel = SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) x 1024 SIMDs)
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def rows(directory):
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            yield from csv.DictReader(f)


def short(name):
    return re.sub(r"\(.*", "", name).replace("void ", "").strip()


def traffic(fdir, wdir, kernel, alg=None):
    def per_launch(directory, counter):
        vals = [float(r["Counter_Value"]) for r in rows(directory) if kernel in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter]
        if not vals:
            raise SystemExit(f"no {counter} rows for {kernel} under {directory}")
        return sum(vals) / len(vals), len(vals)
    f, fn = per_launch(fdir, "FETCH_SIZE")
    w, wn = per_launch(wdir, "WRITE_SIZE")
    out = {"kernel": kernel, "FETCH_SIZE_KB_per_launch": f, "launches_fetch_pass": fn, "WRITE_SIZE_KB_per_launch": w,
           "launches_write_pass": wn, "traffic_bytes_per_launch": int((2 * f + w) * 1024),
           "correction": "FETCH_SIZE x2 (gfx950: 16-B/lane coalesced reads are tallied at half), WRITE_SIZE x1, unit KB"}
    if alg:
        out["algorithmic_bytes_per_launch"] = int(alg)
        out["traffic_over_algorithmic"] = round(out["traffic_bytes_per_launch"] / float(alg), 3)
    print(json.dumps(out, indent=1))


def shapes(fdir, wdir, cands, merge=None):
    def series(directory, counter):
        out = defaultdict(list)
        for r in rows(directory):
            if r.get("Counter_Name") == counter:
                out[short(r.get("Kernel_Name", ""))].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
        return {k: [v for _, v in sorted(vs)] for k, vs in out.items()}
    F, W = series(fdir, "FETCH_SIZE"), series(wdir, "WRITE_SIZE")
    # the inference instantiation with the in-launch output head: "opnet_xcd_forward<true>" until round 5, "<true, false>" since the
    # kernel has a TRAIN template parameter
    fwd = next((k for k in ("opnet_xcd_forward<true, false>", "opnet_xcd_forward<true>") if k in F), "opnet_xcd_forward<true, false>")
    if fwd not in F or len(F[fwd]) != len(W.get(fwd, [])):
        raise SystemExit("the two passes do not hold the same launches of " + fwd)
    compulsory = lambda n: n * 130800 + 8 * 5.68e6      # boxes read + logits + y = 130.8 KB per clip, weights once per XCD
    want_fetch = lambda n: n * 115200 + 8 * 5.68e6
    shape = [min(cands, key=lambda n: abs(want_fetch(n) - 2 * f * 1024)) for f in F[fwd]]
    out = {}
    for kern in (fwd, "opnet_xcd_pack_input", "opnet_xcd_y_poison"):
        if kern not in F or kern not in W:
            continue
        # kernels launched once per forward: the LAST len(shape) launches belong to the head-once forwards, in order
        fs, ws = F[kern][-len(shape):], W[kern][-len(shape):]
        for n in sorted(set(shape)):
            idx = [i for i, s_ in enumerate(shape) if s_ == n and i < len(fs)]
            if not idx:
                continue
            f = sum(fs[i] for i in idx) / len(idx)
            w = sum(ws[i] for i in idx) / len(idx)
            rec = {"bytes_per_launch": int((2 * f + w) * 1024), "FETCH_SIZE_KB": round(f, 1), "WRITE_SIZE_KB": round(w, 1), "launches": len(idx),
                   "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over `bench.py --gpus 1 --steps 20 --warmup 5`; "
                           "FETCH_SIZE x2 (gfx950 correction, MI355X_MICROARCH.md), WRITE_SIZE x1, unit KB"}
            if kern == fwd:
                rec["compulsory_bytes"] = int(compulsory(n))
                rec["over_compulsory"] = round(rec["bytes_per_launch"] / compulsory(n), 3)
                rec["note"] += "; compulsory = 130.8 KB per clip (boxes read, logits, y) + the weights once per XCD"
            out.setdefault(kern.replace("<true, false>", "").replace("<true>", ""), {})[str(n)] = rec
    if merge and os.path.exists(merge):
        old = json.load(open(merge))
        for k, v in old.items():
            if k not in out and ("full-history" in k or "round" in k):
                out[k] = v
    print(json.dumps(out, indent=1))


def mfma(dirs):
    acc = defaultdict(lambda: defaultdict(float))
    for d in dirs:
        for r in rows(d):
            acc[short(r.get("Kernel_Name", ""))][r.get("Counter_Name")] += float(r["Counter_Value"])
    out = {}
    for k, c in sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
        if c.get("GRBM_GUI_ACTIVE", 0) <= 0 or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
            continue
        out[k] = {"SQ_VALU_MFMA_BUSY_CYCLES": int(c["SQ_VALU_MFMA_BUSY_CYCLES"]), "GRBM_GUI_ACTIVE": int(c["GRBM_GUI_ACTIVE"]),
                  "mfma_util": round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)}
    print(json.dumps({"formula": "SQ_VALU_MFMA_BUSY_CYCLES / ((GRBM_GUI_ACTIVE / 8 XCDs) * 1024 SIMDs), summed over all launches of the kernel",
                      "kernels": out}, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "shapes":
        shapes(sys.argv[2], sys.argv[3], [int(x) for x in sys.argv[4].split(",")], sys.argv[5] if len(sys.argv) > 5 else None)
    elif sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4], float(sys.argv[5]) if len(sys.argv) > 5 else None)
    else:
        mfma(sys.argv[2:])
