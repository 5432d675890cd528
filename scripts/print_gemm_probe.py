#!/usr/bin/env python
"""usage: print_gemm_probe.py trace.csv K1 K2 ...  (12 launches per K, in order)"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_gemm" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
Ks = sys.argv[2:]
per = len(rows) // max(len(Ks), 1)
for i, K in enumerate(Ks):
    seg = rows[i * per:(i + 1) * per]
    v = sorted((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in seg[2:])
    print(f"K={K:>6s} grid=({seg[0]['Grid_Size_X']},{seg[0]['Grid_Size_Z']}) n={len(v):3d} median {v[len(v)//2]:7.2f} us  min {v[0]:7.2f} us")
