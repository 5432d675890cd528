#!/bin/bash
# Round 6 (same method as gpu_pmc5.sh, the library that ships in round 6): fabric-side traffic of one steady-state iteration of the fused solvers from PMC counters — two separate passes per solver
# (FETCH_SIZE takes 3 TCC slots, WRITE_SIZE 2 — MI355X_MICROARCH.md, rocprofv3 PMC slots), each with --kernel-trace only.  Iterations
# are cut out of the DISPATCH ORDER (the six launches from one k_wskpl to the next / the six from one k_graw to the next for Neumann),
# not out of per-kernel averages: since round 5 k_wskpc also runs in the once-per-step passes.  Writes gpurun_out/pmc/r06_pmc_traffic.json,
# stamped with the sha256 of the libbhg.so that ran (bench.py replays `roofline.traffic` only on a matching stamp).
set -u
mkdir -p gpurun_out/pmc; export TMPDIR=/tmp
for ALGO in ${ALGOS:-cg neumann}; do
  EXTRA=""; [ $ALGO = neumann ] && EXTRA="--algo neumann --cg-iters 10"
  for C in FETCH_SIZE WRITE_SIZE; do
    cd /tmp && rm -rf /tmp/pmc_${ALGO}_$C && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_${ALGO}_$C -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --settle-ms 0 --reps 1 --cpu-steps 0 --no-kernel-timing --no-slope --no-parity $EXTRA > /tmp/pmc_${ALGO}_$C.log 2>&1; echo "$ALGO $C rc=$?"
  done
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json, collections, hashlib, sys, os
sys.path.insert(0, os.getcwd())
from betty_amd import _native
sha = hashlib.sha256(open(_native.LIB_PATH, "rb").read()).hexdigest()
N = 10034826
def short(n):
    return n.replace("bhg::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
def dispatches(algo, C):
    f = glob.glob(f"/tmp/pmc_{algo}_{C}/*counter_collection.csv")
    if not f:
        return []
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])):
        if r.get("Counter_Name") != C:
            continue
        d = int(r["Dispatch_Id"])
        if d not in per:
            per[d] = [short(r["Kernel_Name"]), 0.0]
        per[d][1] += float(r["Counter_Value"])
    return [per[d] for d in sorted(per)]
out = {"workload_N": N, "lib_sha256": sha, "units": "FETCH_SIZE / WRITE_SIZE are KiB per dispatch; FETCH_SIZE x2 per MI355X_MICROARCH.md (gfx950 counts 128-B "
       "requests at 64 B); iterations cut out of the dispatch order", "per_kernel": {}, "traffic_bytes": {}, "iteration_pattern": {}}
for algo, first, n_launch, key in (("cg", "k_wskpl", 6, "cg_iter_fused"), ("neumann", "k_graw", 6, "neumann_iter_fused")):
    fe, wr = dispatches(algo, "FETCH_SIZE"), dispatches(algo, "WRITE_SIZE")
    if not fe or len(fe) != len(wr):
        print(algo, "passes missing or of different length", len(fe), len(wr)); continue
    rows = [(a[0], 2 * 1024 * a[1], 1024 * b[1]) for a, b in zip(fe, wr)]
    tab = collections.defaultdict(lambda: [0, 0.0, 0.0])
    for n, f_, w_ in rows:
        if n.startswith("k_"):
            tab[n][0] += 1; tab[n][1] += f_; tab[n][2] += w_
    out["per_kernel"][algo] = {k: {"launches": v[0], "fetch_bytes": v[1] / v[0], "write_bytes": v[2] / v[0]} for k, v in sorted(tab.items())}
    idx = [i for i, r in enumerate(rows) if r[0].startswith(first)]
    segs = [rows[a:b] for a, b in zip(idx, idx[1:]) if b - a == n_launch and all(r[0].startswith("k_") for r in rows[a:b])]
    pats = collections.Counter(tuple(r[0] for r in s) for s in segs)
    if not pats:
        print(algo, "no steady-state iteration found"); continue
    pat, cnt = pats.most_common(1)[0]
    good = [s for s in segs if tuple(r[0] for r in s) == pat]
    tot = sum(sum(r[1] + r[2] for r in s) for s in good) / len(good)
    out["traffic_bytes"][key] = tot
    out["iteration_pattern"][key] = {"launches": list(pat), "iterations_averaged": len(good),
                                     "per_launch_bytes": [sum(s[j][1] + s[j][2] for s in good) / len(good) for j in range(n_launch)]}
    steps = sum(1 for r in rows if r[0].startswith("k_cg_init" if algo == "cg" else "k_neumann_init"))
    out["traffic_bytes"][key.replace("iter_fused", "step_total")] = sum(r[1] + r[2] for r in rows if r[0].startswith("k_")) / max(steps, 1)
json.dump(out, open("gpurun_out/pmc/r06_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out["traffic_bytes"], indent=1))
for key, v in out["iteration_pattern"].items():
    print(key, v["iterations_averaged"], "iterations")
    for n, b in zip(v["launches"], v["per_launch_bytes"]):
        print(f"   {n:44s} {b / 1e6:8.2f} MB")
PY
# ---- the batch-norm kernels: bytes per element from the same counters (expected 12 read by k_bn_vjp_stats, 12 read + 8 written by k_bn_vjp_apply)
for C in FETCH_SIZE WRITE_SIZE; do
  cd /tmp && rm -rf /tmp/pmc_bn_$C && timeout 600 rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/pmc_bn_$C -o p -- python $GRAFT_REPO_ROOT/scripts/bench_bn.py > /tmp/pmc_bn_$C.log 2>&1; echo "bn $C rc=$?"
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, json, collections
def dispatches(C):
    f = glob.glob(f"/tmp/pmc_bn_{C}/*counter_collection.csv")
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f[0])) if f else []:
        if r.get("Counter_Name") != C or "k_bn_vjp" not in r["Kernel_Name"]:
            continue
        d = int(r["Dispatch_Id"])
        per.setdefault(d, ["stats" if "stats" in r["Kernel_Name"] else "apply", int(r["Grid_Size"]) if "Grid_Size" in r else 0, 0.0])
        per[d][2] += float(r["Counter_Value"])
    return [per[d] for d in sorted(per)]
fe, wr = dispatches("FETCH_SIZE"), dispatches("WRITE_SIZE")
shapes = [(25, 32, 84, 84), (25, 80, 42, 42), (25, 160, 21, 21), (25, 320, 10, 10), (64, 256, 56, 56), (128, 64, 112, 112)]
out = {"units": "FETCH_SIZE x2 (gfx950: 128-B requests tallied at 64 B) and WRITE_SIZE, KiB per dispatch -> bytes; 53 calls per shape (3 warm-up + 50 timed), two launches per call",
       "expected_bytes_per_element": {"k_bn_vjp_stats": {"read": 12, "write": 0}, "k_bn_vjp_apply": {"read": 12, "write": 8}}, "shapes": []}
if fe and len(fe) == len(wr) == 2 * 53 * len(shapes):
    for i, sh in enumerate(shapes):
        n = sh[0] * sh[1] * sh[2] * sh[3]
        seg_f, seg_w = fe[2 * 53 * i: 2 * 53 * (i + 1)], wr[2 * 53 * i: 2 * 53 * (i + 1)]
        row = {"shape": list(sh), "elements": n}
        for kind in ("stats", "apply"):
            f_ = [2 * 1024 * a[2] for a in seg_f if a[0] == kind]
            w_ = [1024 * a[2] for a in seg_w if a[0] == kind]
            row[f"k_bn_vjp_{kind}"] = {"fetch_bytes_per_element": sum(f_) / len(f_) / n, "write_bytes_per_element": sum(w_) / len(w_) / n}
        out["shapes"].append(row)
        print(sh, {k: {kk: round(vv, 2) for kk, vv in v.items()} for k, v in row.items() if k.startswith("k_")})
else:
    print("bn passes missing or of unexpected length", len(fe), len(wr))
json.dump(out, open("gpurun_out/pmc/r06_pmc_bn_traffic.json", "w"), indent=1)
PY
