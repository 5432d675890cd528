#!/usr/bin/env python
"""Splits a rocprofv3 kernel trace of scripts/cfg3_profile.py into kernel classes and idle time (see that script)."""
import csv
import re
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the timed steps: everything after the last idle gap > 0.5 s
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - rows[i - 1][1] > 500_000_000:
        cut = i
rows = rows[cut:]
span = rows[-1][1] - rows[0][0]
CLASSES = [
    ("libbhg (this library: recurrence, flatten)", r"^(void )?(bhg::)?k_"),
    ("MIOpen / convolution", r"(?i)miopen|conv|igemm|winograd|sp3|gcnasm|direct|Im2Col|Col2Im|implicit|xdlops|naive_conv|SubTensor|transpose_|batched_transpose|kernel_gemm_xdl|ck::|ck_tile"),
    ("batch-norm (native batch_norm kernels)", r"(?i)batch_norm|batchnorm|bn_"),
    ("GEMM (rocBLAS / hipBLASLt / Tensile)", r"(?i)Cijk_|rocblas|gemm|tensile"),
    ("reduce", r"(?i)reduce_kernel|reduce"),
    ("element-wise / copy / fill (ATen)", r"(?i)elementwise|vectorized|unrolled|CatArray|copy|fill|index|where|pow|mul|add|threshold|relu"),
]
tot = defaultdict(lambda: [0, 0])
names = defaultdict(lambda: defaultdict(lambda: [0, 0]))
busy = 0
last_end = rows[0][0]
idle = 0
for s, e, n in rows:
    d = e - s
    busy += d
    if s > last_end:
        idle += s - last_end
    last_end = max(last_end, e)
    for cls, pat in CLASSES:
        if re.search(pat, n):
            break
    else:
        cls = "other"
    tot[cls][0] += d
    tot[cls][1] += 1
    names[cls][n[:110]][0] += d
    names[cls][n[:110]][1] += 1
print(f"timed part of the trace: {len(rows)} kernel launches, span {span / 1e6:.1f} ms, kernel time {busy / 1e6:.1f} ms ({100 * busy / span:.1f} %), "
      f"idle (no kernel running) {idle / 1e6:.1f} ms ({100 * idle / span:.1f} %)")
for cls, (d, c) in sorted(tot.items(), key=lambda kv: -kv[1][0]):
    print(f"  {cls:48s} {d / 1e6:9.1f} ms  {100 * d / span:5.1f} % of span  {c:7d} launches  avg {d / c / 1e3:7.1f} us")
    for n, (dd, cc) in sorted(names[cls].items(), key=lambda kv: -kv[1][0])[:6]:
        print(f"      {dd / 1e6:8.1f} ms {cc:6d} x {dd / cc / 1e3:8.1f} us  {n}")
