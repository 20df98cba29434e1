#!/bin/bash
# Run on the GPU box from the repo root: the driver's command under rocprofv3 --kernel-trace --stats -> gpurun_out/r6_bench_kernel_stats.csv
# and the per-launch durations of the headline kernel (the run holds 640-clip timed launches, 400-clip ones of the other-batch blocks
# and 1024-clip ones of the files block: the stats' average mixes them) -> gpurun_out/r6_bench_xcd_forward_launches.csv
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/bkp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bkp -o p -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --repeats 5 > /tmp/bkp.log 2>&1
cp $(find /tmp/bkp -name "*kernel_stats.csv") $R/gpurun_out/r6_bench_kernel_stats.csv
python - "$(find /tmp/bkp -name '*kernel_trace.csv')" <<'PY' > $R/gpurun_out/r6_bench_xcd_forward_launches.csv
import csv, sys
print("dispatch,kernel,duration_ns")
for r in csv.DictReader(open(sys.argv[1])):
    if "opnet_xcd_forward" in r["Kernel_Name"]:
        print(f'{r["Dispatch_Id"]},"{r["Kernel_Name"].split("(")[0].replace("void ", "")}",{int(r["End_Timestamp"]) - int(r["Start_Timestamp"])}')
PY
grep "^{" /tmp/bkp.log | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench line:', d['value'], 'launch_ms', d['roofline']['launch_ms'], 'frac', d['roofline']['frac'])"
python - <<'PY'
import csv, os, collections
R = os.environ.get("GRAFT_REPO_ROOT", os.getcwd())
d = [int(r["duration_ns"]) for r in csv.DictReader(open(R + "/gpurun_out/r6_bench_xcd_forward_launches.csv")) if "<true, false>" in r["kernel"]]
b = collections.Counter(round(x / 1e6, 1) for x in d)
print("launches of opnet_xcd_forward<true, false> by duration (ms):", sorted(b.items()))
t = [x for x in d if 4.4e6 < x < 5.1e6]
print("640-clip launches:", len(t), "avg ms", sum(t) / len(t) / 1e6 if t else None)
PY
