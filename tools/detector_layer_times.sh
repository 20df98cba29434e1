#!/bin/bash
# per-dispatch durations of the conv kernels of ONE 16-frame detector pass (rocprofv3 kernel trace): which layers cost what
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/detlayers
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R && rocprofv3 --kernel-trace --output-format csv -d $O/tr -o p -- python tools/detector_full_time.py 16 > $O/run.log 2>&1)
cd $R
python - <<'PY'
import csv, glob, os, collections
O = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/detlayers"
rows = []
for path in glob.glob(O + "/tr/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last pass: find the last "preprocess_frame" burst and take everything after its first launch
idx = [i for i, r in enumerate(rows) if "preprocess_frame" in r["Kernel_Name"]]
# passes start with 16 preprocess launches; take the start of the last group
starts = [i for k, i in enumerate(idx) if k == 0 or idx[k - 1] != i - 1]
runs = {s: sum(1 for i in idx if i >= s and i < s + 64 and all(j in idx for j in range(s, i + 1))) for s in starts}
s16 = [s for s in starts if runs[s] >= 16]          # passes that begin with 16 frame preparations (the single-frame breakdown follows them)
begin = s16[-1]
end = min([s for s in starts if s > begin] + [len(rows)])
last = rows[begin:end]
tot = collections.OrderedDict()
total = 0.0
out = []
for r in last:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    total += d
    if "conv2d" in name or "gemm" in name:
        out.append((name, r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("Grid_Size_Z", ""), d))
    tot[name] = tot.get(name, 0.0) + d
with open(O + "/conv_dispatches.txt", "w") as f:
    for k, (n, gx, gy, gz, d) in enumerate(out):
        f.write(f"{k:3d} {n:40s} grid {gx:>8s} {gy:>4s} {gz:>3s}  {d:9.1f} us\n")
    f.write(f"\nall kernels of the pass: {total:.1f} us\n")
    for n, d in sorted(tot.items(), key=lambda kv: -kv[1])[:25]:
        f.write(f"{d:10.1f} us  {n}\n")
print(open(O + "/conv_dispatches.txt").read())
PY
find $O -name "*kernel_trace.csv" -delete
