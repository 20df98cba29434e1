#!/bin/bash
# every kernel of ONE one-frame detector call, in launch order, with its duration and the gap to the previous kernel's end
# (rocprofv3 kernel trace of tools/detector_full_time.py 1) -> gpurun_out/det1/one_frame_kernels.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/det1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
(cd $R && rocprofv3 --kernel-trace --output-format csv -d $O/tr -o p -- python tools/detector_full_time.py 1 > $O/run.log 2>&1)
cd $R
python - <<'PY'
import csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out/det1"
rows = []
for path in glob.glob(O + "/tr/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "preprocess_frame" in r["Kernel_Name"]]
# the stage breakdown at the end of the script is the last call; the timed calls precede it: take the third call from the end
begin, end = idx[-3], idx[-2]
call = rows[begin:end]
t_prev = None
busy = 0.0
with open(O + "/one_frame_kernels.txt", "w") as f:
    for k, r in enumerate(call):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
        gap = (s - t_prev) / 1e3 if t_prev is not None else 0.0
        busy += (e - s) / 1e3
        f.write(f"{k:3d} {name:44s} grid {r.get('Grid_Size_X','?'):>8s} {r.get('Grid_Size_Y',''):>4s} {r.get('Grid_Size_Z',''):>3s}  {(e - s) / 1e3:8.1f} us  gap {gap:6.1f}\n")
        t_prev = e
    span = (int(call[-1]["End_Timestamp"]) - int(call[0]["Start_Timestamp"])) / 1e3
    f.write(f"\n{len(call)} kernels, {busy:.1f} us of kernel time in a span of {span:.1f} us\n")
print(open(O + "/one_frame_kernels.txt").read())
PY
find $O -name "*kernel_trace.csv" -delete
