#!/usr/bin/env python
"""From a rocprofv3 kernel trace of scripts/iter_trace.py: everything that runs OUTSIDE the K loop of one
hypergradient step (after the last fused-output kernel of a step until the next step's k_*_init), with start
offsets so idle gaps are visible."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    return n.replace("bhg::(anonymous namespace)::", "").replace("void ", "").replace("at::native::", "")[:100]
inits = [i for i, r in enumerate(rows) if "k_cg_init" in r["Kernel_Name"] or "k_neumann_init" in r["Kernel_Name"]]
if len(inits) < 2:
    print("need >= 2 steps in the trace"); sys.exit(0)
lo, hi = inits[-2], inits[-1]
seg = rows[lo:hi + 1]
last_loop = max(i for i, r in enumerate(seg) if any(k in r["Kernel_Name"] for k in ("k_outer", "k_cg_alpha", "k_cg_pdir", "k_cg_resident", "k_neumann_step")))
tail = seg[last_loop:]
t0 = int(tail[0]["End_Timestamp"])
busy = 0
print("after the K loop of a step until the next step's k_*_init (mixed VJP, upper backward, next prepare):")
for r in tail[1:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy += e - s
    print(f"{(s - t0) / 1e3:8.1f}us +{(e - s) / 1e3:6.1f}  {short(r['Kernel_Name'])}")
span = (int(tail[-1]["End_Timestamp"]) - t0) / 1e3
print(f"span {span:.1f} us, kernel time {busy / 1e3:.1f} us, idle {span - busy / 1e3:.1f} us, launches {len(tail) - 1}")
