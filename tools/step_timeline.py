"""Print the launch timeline (start us, duration us, kernel, grid) of the last step in a rocprofv3 --kernel-trace CSV: the launches
between the last two launches of the kernel named by argv[2] (default seqx_backward)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
mark = sys.argv[2] if len(sys.argv) > 2 else "seqx_backward"
idx = [i for i, r in enumerate(rows) if mark in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["End_Timestamp"])
for r in rows[a + 1:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {r['Kernel_Name'][:60]:60s} grid {r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']} wg {r['Workgroup_Size_X']}")
