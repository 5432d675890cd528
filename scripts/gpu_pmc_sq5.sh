#!/bin/bash
# SQ counters of the default path, per kernel (one pass, --kernel-trace only): where the waves' cycles go in the launches of an
# iteration — MFMA pipe busy, wave parked (s_waitcnt / barrier), issue stalls, LDS bank conflicts.
set -u
mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_sq -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --settle-ms 0 --reps 1 --cpu-steps 0 --no-kernel-timing --no-slope --no-parity > /tmp/pmc_sq.log 2>&1; echo "sq rc=$?"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json, collections, hashlib, sys, os
sys.path.insert(0, os.getcwd())
from betty_amd import _native
sha = hashlib.sha256(open(_native.LIB_PATH, "rb").read()).hexdigest()
f = glob.glob("/tmp/pmc_sq/*counter_collection.csv")
if not f:
    print(open("/tmp/pmc_sq.log").read()[-2000:]); raise SystemExit("no counter file")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f[0])):
    if "bhg" not in r["Kernel_Name"]:
        continue
    name = r["Kernel_Name"].replace("bhg::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"lib_sha256": sha, "command": "bench.py --steps 3 --warmup 1 (default fully projected CG solver), rocprofv3 --kernel-trace --pmc <8 SQ counters>",
       "units": "sums over all SEs per dispatch, averaged over the dispatches of a kernel; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* in quad-cycles, "
                "SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES in cycles (MI355X_MICROARCH.md)", "per_kernel": {}}
for k in sorted(agg):
    row = {c: sum(v) / len(v) for c, v in agg[k].items()}
    row["dispatches"] = len(next(iter(agg[k].values())))
    wc = row.get("SQ_WAVE_CYCLES", 0.0)
    if wc:
        row["share_wave_parked"] = row.get("SQ_WAIT_ANY", 0.0) / wc
        row["share_issue_stall"] = row.get("SQ_WAIT_INST_ANY", 0.0) / wc
        row["share_issuing"] = row.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
    if row.get("SQ_BUSY_CYCLES"):
        row["mfma_busy_over_sq_busy"] = row.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / row["SQ_BUSY_CYCLES"]
    if row.get("SQ_LDS_IDX_ACTIVE"):
        row["lds_conflict_share"] = row.get("SQ_LDS_BANK_CONFLICT", 0.0) / row["SQ_LDS_IDX_ACTIVE"]
    out["per_kernel"][k] = row
json.dump(out, open("gpurun_out/pmc/r05_pmc_sq_default.json", "w"), indent=1)
for k, row in out["per_kernel"].items():
    if row["dispatches"] >= 30:
        print("%-40s n=%4d parked %.2f stall %.2f issuing %.2f mfma/sqbusy %s lds-conflict %s" % (
            k[:40], row["dispatches"], row.get("share_wave_parked", 0), row.get("share_issue_stall", 0), row.get("share_issuing", 0),
            "%.3f" % row["mfma_busy_over_sq_busy"] if "mfma_busy_over_sq_busy" in row else "-",
            "%.3f" % row["lds_conflict_share"] if "lds_conflict_share" in row else "-"))
PY
