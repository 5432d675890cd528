#!/bin/bash
set -u
export TMPDIR=/tmp
cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python $GRAFT_REPO_ROOT/scripts/hvp_trace.py 5 > /tmp/tr.log 2>&1; echo rc=$?
python - <<'PY'
import csv
rows=list(csv.DictReader(open('/tmp/tr/t_kernel_trace.csv')))
rows=[r for r in rows if 'bhg' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last HVP = last 19 launches
n=int(__import__('os').environ.get('NLAUNCH','15'))
last=rows[-n:]
t0=int(last[0]['Start_Timestamp'])
for r in last:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    name=r['Kernel_Name'].replace('bhg::(anonymous namespace)::','').replace('void ','').split('(')[0]
    print(f"{(s-t0)/1e3:8.1f}us +{(e-s)/1e3:7.1f}us  {name:28s} grid=({r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']}) wg={r['Workgroup_Size_X']} vgpr={r.get('VGPR_Count','')} lds={r.get('LDS_Block_Size','')}")
print("HVP span us", (int(last[-1]['End_Timestamp'])-t0)/1e3)
PY
