#!/bin/bash
# HBM traffic of the recurrence kernel from PMC counters: two separate passes (FETCH_SIZE takes 3 TCC
# slots, WRITE_SIZE 2 — MI355X_MICROARCH.md §rocprofv3 PMC slots), each with --kernel-trace only.
set -u
mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_$C -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-steps 0 --no-kernel-timing "$@" > /tmp/pmc_$C.log 2>&1; echo "$C rc=$?"
  ls /tmp/pmc_$C | head
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json, collections
out={}
for C in ("FETCH_SIZE","WRITE_SIZE"):
    f=glob.glob(f"/tmp/pmc_{C}/*counter_collection.csv")
    if not f: print("no file for",C); continue
    rows=list(csv.DictReader(open(f[0])))
    agg=collections.defaultdict(list)
    for r in rows:
        if r.get("Counter_Name")!=C: continue
        name=r["Kernel_Name"].split("(")[0].replace("bhg::(anonymous namespace)::","").replace("void ","")
        if "bhg" in r["Kernel_Name"]:
            agg[r["Kernel_Name"].replace("bhg::(anonymous namespace)::","").replace("void ","").split("(")[0]].append(float(r["Counter_Value"]))
    out[C]={k:{"n":len(v),"avg":sum(v)/len(v)} for k,v in agg.items()}
print(json.dumps(out,indent=1))
json.dump(out,open("gpurun_out/pmc/pmc_summary.json","w"),indent=1)
PY
