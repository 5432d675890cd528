#!/bin/bash
# Fabric-side traffic of one fused CG-HVP iteration (and of the un-fused recurrence kernel) from PMC counters: two
# separate passes (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2 — MI355X_MICROARCH.md §rocprofv3 PMC slots), each with
# --kernel-trace only.  Writes profiles-ready JSON to gpurun_out/pmc/r04_pmc_traffic.json, stamped with the sha256 of
# the libbhg.so that ran (bench.py replays `traffic` only when the stamp matches the loaded library).
set -u
mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
for ARM in fused; do
  EXTRA=""; [ $ARM = nofuse ] && EXTRA="--no-fuse"
  for C in FETCH_SIZE WRITE_SIZE; do
    cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_${ARM}_$C -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --cpu-steps 0 --no-kernel-timing --no-slope --no-parity $EXTRA > /tmp/pmc_${ARM}_$C.log 2>&1; echo "$ARM $C rc=$?"
  done
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json, collections, hashlib, sys, os
sys.path.insert(0, os.getcwd())
from betty_amd import _native
sha = hashlib.sha256(open(_native.LIB_PATH, "rb").read()).hexdigest()
N = 10034826
def per_kernel(arm, C):
    f = glob.glob(f"/tmp/pmc_{arm}_{C}/*counter_collection.csv")
    if not f:
        return {}
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") != C or "bhg" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].replace("bhg::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[name].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in agg.items()}
out = {"workload_N": N, "lib_sha256": sha, "units": "FETCH_SIZE / WRITE_SIZE are KiB per dispatch; FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 counts 128-B requests at 64 B)",
       "per_kernel": {}, "traffic_bytes": {}}
steps, K = 4, 20   # 1 warm-up + 3 steps
for arm in ("fused",):
    fe, wr = per_kernel(arm, "FETCH_SIZE"), per_kernel(arm, "WRITE_SIZE")
    tab = {}
    for k in sorted(set(fe) | set(wr)):
        n = (fe.get(k) or wr.get(k))[0]
        tab[k] = {"launches": n, "fetch_bytes": 2 * 1024 * (fe.get(k, (0, 0))[1]), "write_bytes": 1024 * (wr.get(k, (0, 0))[1])}
    out["per_kernel"][arm] = tab
    # the steady-state launches of a projected iteration (k_wskpl, k_wskpu, k_head_forward, k_wskpc x2, k_graw; the first iteration's
    # k_wskpl / k_wskpc / k_graw launches carry the same names and are counted with them); everything else is once per solve
    loop = [k for k in tab if k.startswith(("k_wskpc", "k_wskpl", "k_wskpu", "k_graw", "k_pstep")) or k == "k_head_forward<true, 3, true, true>"]
    solves = tab["k_cg_init"]["launches"] if "k_cg_init" in tab else steps
    tot = sum((tab[k]["fetch_bytes"] + tab[k]["write_bytes"]) * tab[k]["launches"] for k in loop) / (solves * K)
    out["traffic_bytes"]["cg_iter_fused"] = tot
    out["traffic_bytes"]["outside_the_iterations_per_step"] = sum((tab[k]["fetch_bytes"] + tab[k]["write_bytes"]) * tab[k]["launches"] for k in tab if k not in loop) / solves
    out["steady_state_kernels"] = loop
json.dump(out, open("gpurun_out/pmc/r04_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out["traffic_bytes"], indent=1))
for arm in out["per_kernel"]:
    print(arm)
    for k, v in out["per_kernel"][arm].items():
        print(f"  {k:40s} n={v['launches']:5d} fetch={v['fetch_bytes']/1e6:8.2f} MB write={v['write_bytes']/1e6:8.2f} MB")
PY
