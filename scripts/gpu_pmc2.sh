#!/bin/bash
set -u
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d /tmp/pm -o p -- python $GRAFT_REPO_ROOT/scripts/hvp_trace.py 3 > /tmp/pm.log 2>&1; echo rc=$?
tail -3 /tmp/pm.log
python - <<'PY'
import csv, glob, collections
f=glob.glob("/tmp/pm/*counter_collection.csv")
rows=list(csv.DictReader(open(f[0])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if 'bhg' not in r['Kernel_Name']: continue
    key=(r['Kernel_Name'].replace('bhg::(anonymous namespace)::','').replace('void ','').split('(')[0], r['Grid_Size'])
    agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    d={c:sum(x)/len(x) for c,x in v.items()}
    wc=d.get('SQ_WAVE_CYCLES',1)
    print(f"{k[0]:24s} grid={k[1]:>8s} wavecyc={wc:12.0f} busy={d.get('SQ_BUSY_CYCLES',0):10.0f} wait_any={d.get('SQ_WAIT_ANY',0)/wc:5.2f} wait_inst={d.get('SQ_WAIT_INST_ANY',0)/wc:5.2f} active={d.get('SQ_ACTIVE_INST_ANY',0)/wc:5.2f} mfma_busy={d.get('SQ_VALU_MFMA_BUSY_CYCLES',0):12.0f} ldsconf={d.get('SQ_LDS_BANK_CONFLICT',0):10.0f} mops={d.get('SQ_INSTS_VALU_MFMA_MOPS_F32',0):12.0f}")
PY
