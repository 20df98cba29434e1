"""Print the dispatches around the last opnet_xcd_forward launch of a rocprofv3 kernel trace (timeline of one launch)."""
import csv, glob, sys
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "opnet_xcd_forward" in r["Kernel_Name"]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if len(sys.argv) > 3:       # the n longest launches instead of the n last ones
    idx = sorted(idx, key=lambda i: int(rows[i]["End_Timestamp"]) - int(rows[i]["Start_Timestamp"]))
for last in sorted(idx[-n:]):
    t0 = int(rows[last]["Start_Timestamp"])
    print("---")
    for r in rows[max(0, last - 8):last + 12]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print(f"{(s - t0) / 1e3:10.1f} us  +{(e - s) / 1e3:9.1f} us  {r['Kernel_Name'][:90]}")
