#!/usr/bin/env python
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if "k_gemm" in r["Kernel_Name"]:
        key = (r["Grid_Size_X"], r["Grid_Size_Z"])
        agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: int(kv[0][0]) * int(kv[0][1])):
    v = sorted(v[2:]) if len(v) > 4 else sorted(v)
    print(f"grid_x={k[0]:>7s} grid_z={k[1]:>3s}  n={len(v):3d}  median {v[len(v)//2]:7.2f} us  min {v[0]:7.2f} us")
