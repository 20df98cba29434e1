#!/usr/bin/env python3
"""Reduce rocprofv3 PMC passes (one counter set per pass, collected with --pmc only - never together with trace domains) to
per-kernel numbers.

    python tools/pmc_reduce.py traffic <fetch_dir> <write_dir> <kernel-substring> [<algorithmic bytes per launch>]
        HBM-side bytes per launch of the kernel: FETCH_SIZE x 2 (MI355X_MICROARCH.md HBM section: on gfx950 the counter
        reports 1/2 of the bytes of 16-B/lane coalesced reads) + WRITE_SIZE, counter unit KB
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
    if sys.argv[1] == "traffic":
        traffic(sys.argv[2], sys.argv[3], sys.argv[4], float(sys.argv[5]) if len(sys.argv) > 5 else None)
    else:
        mfma(sys.argv[2:])
