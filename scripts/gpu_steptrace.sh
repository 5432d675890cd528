#!/bin/bash
export TMPDIR=/tmp
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/st -o t -- python $GRAFT_REPO_ROOT/scripts/step_trace.py > /tmp/st.log 2>&1; echo rc=$?
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/st/t_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# find last step: from last k_cg_init backwards to previous end... take kernels after the second-to-last k_cg_resident group
idx=[i for i,r in enumerate(rows) if 'k_cg_init' in r['Kernel_Name']]
start=idx[-1]
# walk back to include prepare kernels: everything after the previous step's last kernel (mixed vjp) -> find previous k_cg_resident last index
prev_res=[i for i,r in enumerate(rows[:start]) if 'k_cg_resident' in r['Kernel_Name']]
lo=prev_res[-1]+1 if prev_res else 0
seg=rows[lo:]
t0=int(seg[0]['Start_Timestamp'])
tot=0
print("kernels between the previous step's last CG iteration and the end of this step (non-loop ones listed)")
for r in seg:
    n=r['Kernel_Name']
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    inloop=('k_gemm' in n or 'k_outer' in n or 'k_reduce' in n or 'k_head' in n or 'k_bias' in n or 'k_cg_resident' in n)
    if not inloop:
        tot+=e-s
        print(f"{(s-t0)/1e3:9.1f}us +{(e-s)/1e3:6.1f}  {n[:110]}")
print("non-loop kernel time us:", tot/1e3, " span us:", (int(seg[-1]['End_Timestamp'])-t0)/1e3)
PY
