#!/bin/bash
# Per-kernel matrix-pipe busy fraction and wave stall breakdown of the fused CG iteration (separate PMC passes,
# --kernel-trace only).  -> gpurun_out/pmc/r02_mfma_busy.json
set -u
mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
P1="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
P2="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
i=0
for P in "$P1" "$P2"; do
  i=$((i+1))
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $P --output-format csv -d /tmp/sq_$i -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-steps 0 --no-kernel-timing > /tmp/sq_$i.log 2>&1; echo "pass $i rc=$?"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2):
    f = glob.glob(f"/tmp/sq_{i}/*counter_collection.csv")
    if not f:
        print("no counter file for pass", i); continue
    for r in csv.DictReader(open(f[0])):
        if "bhg" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].replace("bhg::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        key = f"{name} grid={r.get('Grid_Size', r.get('Grid_Size_X', '?'))}"
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, cs in sorted(agg.items()):
    m = {c: sum(v) / len(v) for c, v in cs.items()}
    m["launches"] = max(len(v) for v in cs.values())
    if "SQ_VALU_MFMA_BUSY_CYCLES" in m and m.get("GRBM_GUI_ACTIVE"):
        m["mfma_busy_frac_of_1024_simds"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * m["GRBM_GUI_ACTIVE"])
    if m.get("SQ_WAVE_CYCLES"):
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in m:
                m[c + "_frac_of_wave_cycles"] = m[c] / m["SQ_WAVE_CYCLES"]
    out[k] = m
json.dump(out, open("gpurun_out/pmc/r02_mfma_busy.json", "w"), indent=1)
for k, m in out.items():
    print(f"{k[:60]:60s} n={m['launches']:4d} mfma_busy={m.get('mfma_busy_frac_of_1024_simds', float('nan')):.3f} wait_any={m.get('SQ_WAIT_ANY_frac_of_wave_cycles', float('nan')):.2f} wait_inst={m.get('SQ_WAIT_INST_ANY_frac_of_wave_cycles', float('nan')):.2f} active={m.get('SQ_ACTIVE_INST_ANY_frac_of_wave_cycles', float('nan')):.2f} lds_conf={m.get('SQ_LDS_BANK_CONFLICT', 0):.0f}")
PY
