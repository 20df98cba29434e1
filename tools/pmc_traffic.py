"""Reduce two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE - collected separately, as MI355X_MICROARCH.md prescribes) over
`bench.py --steps 4 --warmup 1 --no-cpu-baseline` to HBM-side bytes per opnet_step launch -> profiles/r1_pmc_b<B>.json.

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python $REPO/bench.py --steps 4 --warmup 1 --no-cpu-baseline
    python $REPO/tools/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w 32 > $REPO/gpurun_out/r1_pmc_b32.json
"""
import csv
import glob
import json
import os
import sys

W_BYTES = 5_684_224
STATE_BYTES_PER_CLIP = 12_672


def per_launch(directory, counter, kernel="opnet_step"):
    vals = []
    for path in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if kernel in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                    vals.append(float(row["Counter_Value"]))
    if not vals:
        raise SystemExit(f"no {counter} rows for {kernel} under {directory}")
    vals.sort()
    steady = vals[len(vals) // 2]                       # median launch: the T+3 launches of a forward include 3 ramp steps
    return steady, sum(vals) / len(vals), len(vals)


def main():
    fdir, wdir, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
    f_med, f_mean, f_n = per_launch(fdir, "FETCH_SIZE")
    w_med, w_mean, w_n = per_launch(wdir, "WRITE_SIZE")
    out = {
        "command": "rocprofv3 --pmc <COUNTER> --output-format csv -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline (one pass per counter; tools/pmc_traffic.py)",
        "kernel": "opnet_step", "batch": batch, "frames": 300,
        "correction": "MI355X_MICROARCH.md HBM section: on gfx950 FETCH_SIZE reports 1/2 of the bytes of 16-B/lane coalesced reads (all opnet_step loads are float4) -> x2; WRITE_SIZE x1; counter unit KB",
        "FETCH_SIZE_KB_per_launch_median": f_med, "FETCH_SIZE_KB_per_launch_mean_all": f_mean, "FETCH_SIZE_launches": f_n,
        "WRITE_SIZE_KB_per_launch_median": w_med, "WRITE_SIZE_KB_per_launch_mean_all": w_mean, "WRITE_SIZE_launches": w_n,
        "traffic_bytes_per_launch": int((2 * f_mean + w_mean) * 1024),
        "algorithmic_bytes_per_launch": int((W_BYTES + batch * STATE_BYTES_PER_CLIP) * 300 / 303),
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
