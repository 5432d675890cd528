#!/bin/bash
# PMC traffic of one iteration of the projected Neumann solver (BASELINE cfg 2's own algorithm, K = 10): two separate passes,
# merged into the round's traffic file under "neumann_iter_fused" (same stamp rule: the sha256 of the library that ran).
set -u
mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_neu_$C -o p -- python $GRAFT_REPO_ROOT/bench.py --algo neumann --cg-iters 10 --steps 3 --warmup 1 --cpu-steps 0 --no-kernel-timing --no-slope > /tmp/pmc_neu_$C.log 2>&1; echo "neumann $C rc=$?"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json, collections, hashlib, sys, os
sys.path.insert(0, os.getcwd())
from betty_amd import _native
sha = hashlib.sha256(open(_native.LIB_PATH, "rb").read()).hexdigest()
src = "gpurun_out/pmc/r03_pmc_traffic.json" if os.path.exists("gpurun_out/pmc/r03_pmc_traffic.json") else "profiles/r03_pmc_traffic.json"
base = json.load(open(src))   # (after gpu_pmc3.sh in the same call: the fresh file; on its own: the committed one)
assert base["lib_sha256"] == sha, "the committed traffic file belongs to another library"
def per_kernel(C):
    f = glob.glob(f"/tmp/pmc_neu_{C}/*counter_collection.csv")
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") != C or "bhg" not in r["Kernel_Name"]:
            continue
        name = r["Kernel_Name"].replace("bhg::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[name].append(float(r["Counter_Value"]))
    return {k: (len(v), sum(v) / len(v)) for k, v in agg.items()}
fe, wr = per_kernel("FETCH_SIZE"), per_kernel("WRITE_SIZE")
tab = {}
for k in sorted(set(fe) | set(wr)):
    n = (fe.get(k) or wr.get(k))[0]
    tab[k] = {"launches": n, "fetch_bytes": 2 * 1024 * (fe.get(k, (0, 0))[1]), "write_bytes": 1024 * (wr.get(k, (0, 0))[1])}
steps, K = 4, 10
loop = [k for k in tab if any(s in k for s in ("k_gemm", "k_hoist", "k_wsk_group", "k_proj_", "k_reduce_mask", "k_head_forward<true", "k_outer", "k_bias", "k_head_outer"))]
tot = sum((tab[k]["fetch_bytes"] + tab[k]["write_bytes"]) * tab[k]["launches"] for k in loop) / (steps * K)
base["per_kernel"]["neumann_fused"] = tab
base["traffic_bytes"]["neumann_iter_fused"] = tot
json.dump(base, open("gpurun_out/pmc/r03_pmc_traffic.json", "w"), indent=1)
print("neumann_iter_fused", tot / 1e6, "MB")
for k, v in tab.items():
    print(f"  {k:40s} n={v['launches']:5d} fetch={v['fetch_bytes']/1e6:8.2f} MB write={v['write_bytes']/1e6:8.2f} MB")
PY
timeout 300 python bench.py --algo neumann --cg-iters 10 --cpu-steps 0 > gpurun_out/bench_neumann_fused.json 2> gpurun_out/bench_neumann_fused.err; tail -c 300 gpurun_out/bench_neumann_fused.json
