#!/bin/bash
# usage: tools/prof_stats.sh <outdir-under-gpurun_out> <cmd...>   (run on the GPU box from the repo root)
# rocprofv3 kernel trace + stats of <cmd>; prints the top kernels; the CSV stays under gpurun_out/<outdir>/
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o p -- "$@" > $OUT/run.log 2>&1) || { tail -20 $OUT/run.log; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows[:14]:
    print(f'{r["Name"][:80]:80s} calls {r["Calls"]:>6s} total_us {float(r["TotalDurationNs"]) / 1e3:10.1f} avg_us {float(r["AverageNs"]) / 1e3:9.2f} {float(r["Percentage"]):5.1f} %')
PY
