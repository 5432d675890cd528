#!/usr/bin/env python
"""Print the per-dispatch timeline of ONE solver iteration from a rocprofv3 kernel trace CSV.
usage: print_iter_timeline.py <kernel_trace.csv> [marker-kernel-substring]
The iteration shown is the span between the last two launches of the marker kernel (default k_cg_pdir)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "k_cg_pdir"
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if len(marks) < 3:
    print("marker not found often enough:", marker, len(marks)); sys.exit(0)
a, b = marks[-3], marks[-2]
sel = rows[a:b + 1]
t0 = int(sel[0]["End_Timestamp"])
def short(n):
    return n.replace("bhg::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:34]
for r in sel[1:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", r.get("Stream_Id", "?"))
    print(f"{(s - t0) / 1e3:8.1f}us +{(e - s) / 1e3:7.1f}us q={q:>3s} {short(r['Kernel_Name']):34s} grid=({r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']})")
print("iteration span us", (int(sel[-1]["End_Timestamp"]) - t0) / 1e3)
