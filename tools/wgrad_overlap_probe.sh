#!/bin/bash
# would hiding the weight gradients under the reverse recurrence pay?  The training step as it is, and
# with OPNET_WGRAD_OVERLAP_PROBE=1 (weight-gradient launches on a side stream next to opnet_xcd4_backward - timing only, wrong gradients):
# step time from bench.py and the kernels' own durations / overlap from a rocprofv3 kernel trace.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/wgrad_overlap
mkdir -p $O
cd $R
for p in 0 1; do
  OPNET_WGRAD_OVERLAP_PROBE=$p python bench.py --mode train --steps 20 --warmup 5 --loss l1 --no-cpu-baseline --repeats 5 > $O/bench_$p.json 2> $O/bench_$p.err
  python -c "import json;d=json.load(open('$O/bench_$p.json'));print('probe=$p ms_per_step', d['ms_per_step'], 'final_loss', d['final_loss'])"
done
cd /tmp && export TMPDIR=/tmp
for p in 0 1; do
  (cd $R && OPNET_WGRAD_OVERLAP_PROBE=$p rocprofv3 --kernel-trace --output-format csv -d $O/tr$p -o p -- python bench.py --mode train --steps 10 --warmup 3 --loss l1 --no-cpu-baseline --repeats 1 > $O/tr$p.log 2>&1)
done
cd $R
python - <<'PY'
import csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/wgrad_overlap"
out = []
for p in (0, 1):
    rows = []
    for path in glob.glob(f"{O}/tr{p}/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last training step: from the last opnet_xcd4_forward to the end
    fw = [i for i, r in enumerate(rows) if "opnet_xcd4_forward" in r["Kernel_Name"]]
    step = rows[fw[-2]:fw[-1]] if len(fw) >= 2 else rows[fw[-1]:]
    t0 = int(step[0]["Start_Timestamp"])
    out.append(f"OPNET_WGRAD_OVERLAP_PROBE={p}: one training step (the one before last), kernel start / end in us from the forward's start")
    for r in step:
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a, b = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        if b - a > 3:
            out.append(f"   {a:9.1f} {b:9.1f}  ({b - a:8.1f} us)  {n}")
    out.append(f"   step span {(int(step[-1]['End_Timestamp']) - t0) / 1e3:.1f} us")
open(O + "/timeline.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY
find $O -name "*kernel_trace.csv" -delete
